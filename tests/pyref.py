"""Slow, independent pure-Python re-derivation of the reference octree build for tiny clouds.

Different code path from the C++ oracle: explicit per-node python lists of (x,y,z,rgb) tuples, python floats,
math.fma-free decode via exact rational arithmetic (fractions) to emulate the single-rounding FMA.
Follows /root/reference src/octree/generation.rs:58-253 + codec.rs:102-139 + node.rs:34-42,157-172.
"""
import math
import struct
from fractions import Fraction

U8, U16, F32, F64 = 1, 2, 3, 4
BPC = {U8: 1, U16: 2, F32: 4, F64: 8}


def _fma(a, b, c):
    """round_to_nearest_even(a*b + c) computed exactly."""
    if any(math.isnan(v) or math.isinf(v) for v in (a, b, c)):
        return a * b + c
    exact = Fraction(a) * Fraction(b) + Fraction(c)
    if exact == 0:
        return a * b + c  # sign of zero as IEEE would give for finite operands
    return float(exact)  # Fraction -> float is correctly rounded (RNE)


def _f32(v):
    return struct.unpack("<f", struct.pack("<f", v))[0]


def position_encoding(edge, res):
    l = math.log2(edge / res)
    c = 0 if not (l > 0) else min(int(l), 4294967295)
    bits = (c + 1) & 0xFFFFFFFF
    if bits <= 8:
        return U8
    if bits <= 16:
        return U16
    if bits <= 24:
        return F32
    return F64


def clamp01(v):
    if v < 0.0:
        return 0.0
    elif v > 1.0:
        return 1.0
    return v


def encode(enc, v, mn, edge):
    t = clamp01((v - mn) / edge)
    if enc == U8:
        s = 255.0 * t
        return 0 if not (s > 0) else min(int(s), 255)
    if enc == U16:
        s = 65535.0 * t
        return 0 if not (s > 0) else min(int(s), 65535)
    if enc == F32:
        return _f32(t)
    return t


def decode(enc, raw, mn, edge):
    if enc == U8:
        return _fma(raw / 255.0, edge, mn)
    if enc == U16:
        return _fma(raw / 65535.0, edge, mn)
    return _fma(float(raw), edge, mn)


def to_bytes(enc, raw):
    if enc == U8:
        return struct.pack("<B", raw)
    if enc == U16:
        return struct.pack("<H", raw)
    if enc == F32:
        return struct.pack("<f", raw)
    return struct.pack("<d", raw)


def cube_of(name, root_min, root_edge):
    mn = list(root_min)
    edge = root_edge
    for ch in name[1:]:
        edge /= 2.0
        d = int(ch)
        mn[0] += float((d >> 2) & 1) * edge
        mn[1] += float((d >> 1) & 1) * edge
        mn[2] += float(d & 1) * edge
    return mn, edge


def build(points, colors, bmin, bmax, res, max_points):
    """points: list of (x,y,z); colors: list of (r,g,b). Returns {name: (num_points, enc, xyz bytes, rgb bytes)}."""
    root_edge = max(max(bmax[0] - bmin[0], bmax[1] - bmin[1]), bmax[2] - bmin[2])
    root_min = list(bmin)
    files = {}  # name -> list of (raw codes tuple, color)
    leaves = []

    def node_stream(name):
        mn, edge = cube_of(name, root_min, root_edge)
        enc = position_encoding(edge, res)
        return [((decode(enc, c[0], mn[0], edge), decode(enc, c[1], mn[1], edge), decode(enc, c[2], mn[2], edge)), col)
                for c, col in files[name]]

    def write(name, pts):
        mn, edge = cube_of(name, root_min, root_edge)
        enc = position_encoding(edge, res)
        files.setdefault(name, []).extend(
            ((encode(enc, p[0], mn[0], edge), encode(enc, p[1], mn[1], edge), encode(enc, p[2], mn[2], edge)), col)
            for p, col in pts)

    def split_node(name, stream):
        mn, edge = cube_of(name, root_min, root_edge)
        ctr = [(mn[a] + (mn[a] + edge)) / 2.0 for a in range(3)]
        kids = {}
        for p, col in stream:
            d = (int(p[0] > ctr[0]) << 2) | (int(p[1] > ctr[1]) << 1) | int(p[2] > ctr[2])
            kids.setdefault(d, []).append((p, col))
        files.pop(name, None)
        for d in sorted(kids):
            child = name + str(d)
            files[child] = []
            write(child, kids[d])
            _, cedge = cube_of(child, root_min, root_edge)
            if len(kids[d]) > max_points and cedge > res:
                split_node(child, node_stream(child))
            else:
                leaves.append(child)

    split_node("r", list(zip(points, colors)))
    finished = {}
    todo = list(leaves)
    deepest = max((len(n) - 1 for n in todo), default=0)
    for level in range(deepest, 0, -1):
        now = [n for n in todo if len(n) - 1 == level]
        todo = [n for n in todo if len(n) - 1 != level]
        parents = sorted(set(n[:-1] for n in now))
        for parent in parents:
            files[parent] = []
            for d in range(8):
                child = parent + str(d)
                if child not in files:
                    continue
                pts = node_stream(child)
                files[child] = []
                write(parent, [pt for i, pt in enumerate(pts) if i % 8 == 0])
                write(child, [pt for i, pt in enumerate(pts) if i % 8 != 0])
                finished[child] = len(files[child])
                if not files[child]:
                    del files[child]
            if parent == "r":
                finished["r"] = len(files["r"])
        todo.extend(parents)
    out = {}
    for name, n in finished.items():
        _, edge = cube_of(name, root_min, root_edge)
        enc = position_encoding(edge, res)
        recs = files.get(name, [])
        xyz = b"".join(to_bytes(enc, c[0]) + to_bytes(enc, c[1]) + to_bytes(enc, c[2]) for c, _ in recs)
        rgb = b"".join(bytes(col) for _, col in recs)
        out[name] = (n, enc, xyz, rgb)
    return out
