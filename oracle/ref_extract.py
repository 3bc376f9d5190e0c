"""Build-container helper: pull individual FUNCTIONS out of reference scripts whose module import needs
packages that are absent here (cv2, skimage, torchvision, wandb, ...). The function bodies used by the hot
path (`compute_attention`, `sliding_window`, `concat_crops`, `blend_images_*`) only need numpy / torch / PIL,
so they are compiled from the reference file's AST and executed unmodified. Used only by
oracle/make_golden.py; nothing of the reference's text is stored in this repository."""
import ast

import numpy as np
import torch
import torch.nn as nn


def load_functions(path, names):
    tree = ast.parse(open(path).read(), filename=path)
    picked = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    missing = set(names) - {n.name for n in picked}
    if missing:
        raise KeyError(f"{path}: functions {sorted(missing)} not found")
    ns = {"np": np, "torch": torch, "nn": nn}
    exec(compile(ast.Module(body=picked, type_ignores=[]), path, "exec"), ns)
    return {n: ns[n] for n in names}


def load_classes(path, names, namespace):
    """Same for CLASS definitions (model.py imports timm, absent here; the classes on the path only need torch
    and the reference's own VisionTransformer / trunc_normal_, which the caller supplies in `namespace`)."""
    tree = ast.parse(open(path).read(), filename=path)
    picked = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name in names]
    missing = set(names) - {n.name for n in picked}
    if missing:
        raise KeyError(f"{path}: classes {sorted(missing)} not found")
    ns = dict(namespace)
    exec(compile(ast.Module(body=picked, type_ignores=[]), path, "exec"), ns)
    return {n: ns[n] for n in names}
