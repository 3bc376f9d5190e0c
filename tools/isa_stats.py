#!/usr/bin/env python3
"""Development tool: instruction mix / register / LDS summary of kernels in a hipcc -S listing.
    hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S file.hip -o /tmp/x.s
    python tools/isa_stats.py /tmp/x.s <substring of the mangled name> [...]"""
import re
import sys


def main():
    s = open(sys.argv[1]).read()
    pats = sys.argv[2:]
    for m in re.finditer(r'^(_Z\S+):\s*; @\S+\n', s, re.M):
        name = m.group(1)
        if pats and not any(p in name for p in pats):
            continue
        end = s.index('.end_amdhsa_kernel', m.end()) if '.end_amdhsa_kernel' in s[m.end():] else len(s)
        body = s[m.end():end]
        if 's_endpgm' not in body:
            continue
        code = body[:body.rindex('s_endpgm')]

        def cnt(p):
            return len(re.findall(p, code))
        meta = {}
        for key in ('.amdhsa_next_free_vgpr', '.amdhsa_accum_offset', '.amdhsa_group_segment_fixed_size',
                    '.amdhsa_private_segment_fixed_size', '.amdhsa_next_free_sgpr'):
            mk = re.search(re.escape(key) + r'\s+(\S+)', body)
            meta[key.replace('.amdhsa_', '')] = mk.group(1) if mk else None
        print(name[:150])
        pats_ = dict(mfma=r'v_mfma', lds_dma=r'(buffer|global)_load_\S+.* lds', vmem_ld=r'(buffer|global)_load',
                     vmem_st=r'(buffer|global)_store', ds_read=r'ds_read', ds_write=r'ds_write', barrier=r's_barrier',
                     valu=r'\n\s+v_(?!mfma)', salu=r'\n\s+s_', scratch='scratch_')
        print("  " + "  ".join(f"{k} {cnt(v)}" for k, v in pats_.items()))
        print("  vmcnt waits:", re.findall(r's_waitcnt vmcnt\((\d+)\)', code)[:60])
        print("  ", meta)


if __name__ == "__main__":
    main()
