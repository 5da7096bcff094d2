#!/usr/bin/env python3
"""Generate the constant tables of the Corintho engine from first principles.

The reference keeps four constant tables in corintho_ai/cpp/include/util.h:
  line_breakers[102]   (util.h:85-290)   96-bit move masks, one per (line, top piece)
  gamma_samples[1024]  (util.h:293-638)  Gamma(0.3,1) bucket means for Dirichlet noise
  space_symmetries[8][16], move_symmetries[8][96] (util.h:641-702)

This script does not read the reference.  It re-derives every table from the
rules of the game (see DESIGN.md "Tables") and writes them as C initialiser
macros to
  corintho_ai_amd/csrc/tables.inc   (product, HIP device constants)
  oracle/tables.inc                 (CPU oracle)
`tests/test_tables.py` checks the generated values against digests recorded in
tests/golden/tables.json and, when /root/reference is mounted, against the
reference header parsed as text.

Derivation rules
----------------
line breakers.  A "line" is 3 or 4 collinear spaces whose top pieces are equal
(t = 0 base, 1 column, 2 capital).  The mover must break it or extend it.  The
mask holds every move that can possibly do so:
  * place piece t+1 on an effective line space (t < 2);
  * place piece t on the extension space of a 3-line;
  * move a stack onto an effective line space from a neighbour outside the line
    (only t < 2: nothing can be put on a capital);
  * move the stack off an effective line space to a neighbour outside the line
    (only t >= 1: a bare base cannot move);
  * move a stack onto the extension space from a neighbour outside the line
    (only t >= 1: a move can never leave a base on top).
"Effective" spaces are all three spaces of a 3-line and the two MIDDLE spaces of
a 4-line (changing an end of a 4-line leaves a 3-line).
Three irregularities of the reference table are reproduced on purpose, because
they decide which moves are legal (they are listed in QUIRKS below).

gamma samples.  Bucket i is the conditional mean of X ~ Gamma(0.3, 1) on the
i-th of 1024 equiprobable quantile intervals: 1024 * integral of x f(x) dx,
evaluated with scipy.integrate.quad, rounded to float32.

symmetries.  Eight dihedral maps of the 4x4 board in a fixed order; a move id
maps by mapping its spaces.
"""
import argparse
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# ---------------------------------------------------------------- move codec
def enc_place(r, c, p):
    return 48 + p * 16 + r * 4 + c


def enc_move(a, b):
    (r0, c0), (r1, c1) = a, b
    if c0 < c1:
        return r0 * 3 + c0  # right
    if r0 < r1:
        return 12 + r0 * 4 + c0  # down
    if c0 > c1:
        return 24 + r0 * 3 + (c0 - 1)  # left
    return 36 + (r0 - 1) * 4 + c0  # up


def dec_move(i):
    """-> ('P', piece, None, (r,c)) or ('M', None, from, to)"""
    if i >= 48:
        return ("P", (i - 48) // 16, None, ((i % 16) // 4, i % 4))
    if i < 12:
        return ("M", None, (i // 3, i % 3), (i // 3, i % 3 + 1))
    if i < 24:
        return ("M", None, ((i - 12) // 4, i % 4), ((i - 12) // 4 + 1, i % 4))
    if i < 36:
        return ("M", None, ((i - 24) // 3, i % 3 + 1), ((i - 24) // 3, i % 3))
    return ("M", None, ((i - 36) // 4 + 1, i % 4), ((i - 36) // 4, i % 4))


def nbrs(s):
    r, c = s
    for dr, dc in ((0, 1), (1, 0), (0, -1), (-1, 0)):
        if 0 <= r + dr < 4 and 0 <= c + dc < 4:
            yield (r + dr, c + dc)


# ------------------------------------------------------------- line breakers
def line_list():
    """102 (cells, extension, top) triples in the reference's index order:
    idx = type*12 + i*3 + t for RL,RR,RB,CU,CD,CB; 72 + d*3 + t for the ten
    diagonal kinds D0U,D0D,D0B,D1U,D1D,D1B,S0..S3."""
    out = []
    for typ in range(6):
        col = typ >= 3
        kind = typ % 3  # 0: triple {0,1,2} (ext 3); 1: triple {1,2,3} (ext 0); 2: all four
        for i in range(4):
            sp = (lambda k, i=i: (k, i)) if col else (lambda k, i=i: (i, k))
            if kind == 0:
                cells, ext = [sp(0), sp(1), sp(2)], sp(3)
            elif kind == 1:
                cells, ext = [sp(1), sp(2), sp(3)], sp(0)
            else:
                cells, ext = [sp(0), sp(1), sp(2), sp(3)], None
            for t in range(3):
                out.append((cells, ext, t))
    d0 = [(0, 0), (1, 1), (2, 2), (3, 3)]
    d1 = [(0, 3), (1, 2), (2, 1), (3, 0)]
    for d in (d0, d1):
        for cells, ext in ((d[:3], d[3]), (d[1:], d[0]), (d, None)):
            for t in range(3):
                out.append((cells, ext, t))
    for cells in (
        [(2, 0), (1, 1), (0, 2)],
        [(0, 1), (1, 2), (2, 3)],
        [(1, 3), (2, 2), (3, 1)],
        [(3, 2), (2, 1), (1, 0)],
    ):
        for t in range(3):
            out.append((cells, None, t))
    assert len(out) == 102
    return out


def rule_mask(cells, ext, t):
    m = 0
    inline = set(cells)
    eff = cells[1:3] if len(cells) == 4 else cells
    if t < 2:
        for (r, c) in eff:
            m |= 1 << enc_place(r, c, t + 1)
    if ext is not None:
        m |= 1 << enc_place(ext[0], ext[1], t)
    for b in eff:
        for a in nbrs(b):
            if a in inline:
                continue
            if t < 2:
                m |= 1 << enc_move(a, b)
            if t >= 1:
                m |= 1 << enc_move(b, a)
    if ext is not None and t >= 1:
        for a in nbrs(ext):
            if a not in inline:
                m |= 1 << enc_move(a, ext)
    return m


# Irregular entries of the reference table (util.h:85-290), kept because they
# change which moves are legal.  (line index range, move the rule gives, move the
# reference has instead.)
QUIRKS = [
    # RL i=2 t=1: "up from (3,3) into the extension (2,3)" is stored as "left from (3,3)".
    (range(7, 8), enc_move((3, 3), (2, 3)), enc_move((3, 3), (3, 2))),
    # main-diagonal lines: "down from (1,2) onto (2,2)" is stored one row too high.
    (range(72, 81), enc_move((1, 2), (2, 2)), enc_move((0, 2), (1, 2))),
    # anti-diagonal lines: "down from (1,1) onto (2,1)" is stored one row too high.
    (range(81, 90), enc_move((1, 1), (2, 1)), enc_move((0, 1), (1, 1))),
]


def line_breakers():
    masks = [rule_mask(*l) for l in line_list()]
    for rng, want, have in QUIRKS:
        for idx in rng:
            if masks[idx] >> want & 1:
                masks[idx] = (masks[idx] & ~(1 << want)) | (1 << have)
    return masks


# -------------------------------------------------------------------- gamma
def gamma_samples():
    from scipy.integrate import quad
    from scipy.stats import gamma
    import warnings

    warnings.simplefilter("ignore")
    a = 0.3
    edges = gamma.ppf(np.arange(1025) / 1024.0, a)
    vals = [1024.0 * quad(lambda x: x * gamma.pdf(x, a), edges[i], edges[i + 1])[0] for i in range(1024)]
    return np.asarray(vals, dtype=np.float64).astype(np.float32)


# --------------------------------------------------------------- symmetries
# out[(r,c)] = in[src(r,c)], in the reference's order (util.h:641-650).
SPACE_MAPS = [
    lambda r, c: (r, c),          # identity
    lambda r, c: (r, 3 - c),      # mirror left-right
    lambda r, c: (3 - c, r),      # quarter turn
    lambda r, c: (c, r),          # transpose
    lambda r, c: (3 - r, 3 - c),  # half turn
    lambda r, c: (3 - r, c),      # mirror top-bottom
    lambda r, c: (c, 3 - r),      # quarter turn the other way
    lambda r, c: (3 - c, 3 - r),  # anti-transpose
]


# The reference's MOVE table for the two quarter turns (k = 2, 6) is built from
# the inverse space map, i.e. its policy rows are turned the opposite way to its
# board rows (util.h:653-702 vs :641-650).  Training samples must equal the
# reference's, so the move table uses MOVE_MAP_OF[k] instead of k.
MOVE_MAP_OF = [0, 1, 6, 3, 4, 5, 2, 7]


def symmetries():
    space = np.zeros((8, 16), dtype=np.int32)
    move = np.zeros((8, 96), dtype=np.int32)
    for k, f in enumerate(SPACE_MAPS):
        for r in range(4):
            for c in range(4):
                rr, cc = f(r, c)
                space[k, r * 4 + c] = rr * 4 + cc
    for k in range(8):
        f = SPACE_MAPS[MOVE_MAP_OF[k]]
        for j in range(96):
            kind, p, a, b = dec_move(j)
            if kind == "P":
                rr, cc = f(*b)
                move[k, j] = enc_place(rr, cc, p)
            else:
                move[k, j] = enc_move(f(*a), f(*b))
    return space, move


# ------------------------------------------------------------------ emitters
def words96(m):
    return [(m >> (32 * w)) & 0xFFFFFFFF for w in range(3)]


def build_all():
    lb = line_breakers()
    gm = gamma_samples()
    sp, mv = symmetries()
    return lb, gm, sp, mv


def digests(lb, gm, sp, mv):
    lbw = np.asarray([words96(m) for m in lb], dtype=np.uint32)
    return {
        "line_breakers_u32x3_le": hashlib.sha256(lbw.astype("<u4").tobytes()).hexdigest(),
        "gamma_samples_f32_le": hashlib.sha256(gm.astype("<f4").tobytes()).hexdigest(),
        "space_symmetries_i32_le": hashlib.sha256(sp.astype("<i4").tobytes()).hexdigest(),
        "move_symmetries_i32_le": hashlib.sha256(mv.astype("<i4").tobytes()).hexdigest(),
    }


def emit(lb, gm, sp, mv):
    o = []
    o.append("/* GENERATED by tools/gen_tables.py -- do not edit.  See that script for the")
    o.append(" * derivation of every table and DESIGN.md (Tables) for the reference lines")
    o.append(" * (corintho_ai/cpp/include/util.h:85-702) whose VALUES these must equal. */")
    o.append("#define CO_NUM_LINES 102")
    o.append("#define CO_NUM_GAMMA 1024")
    o.append("/* bit m of the 96-bit mask = word m/32, bit m%32 */")
    o.append("#define CO_LINE_BREAKERS_INIT { \\")
    for m in lb:
        w = words96(m)
        o.append("  {0x%08xu, 0x%08xu, 0x%08xu}, \\" % tuple(w))
    o.append("}")
    o.append("/* float32 bit patterns */")
    o.append("#define CO_GAMMA_BITS_INIT { \\")
    bits = gm.view(np.uint32)
    for i in range(0, 1024, 8):
        o.append("  " + ", ".join("0x%08xu" % b for b in bits[i : i + 8]) + ", \\")
    o.append("}")
    o.append("#define CO_SPACE_SYM_INIT { \\")
    for k in range(8):
        o.append("  {" + ", ".join(str(int(x)) for x in sp[k]) + "}, \\")
    o.append("}")
    o.append("#define CO_MOVE_SYM_INIT { \\")
    for k in range(8):
        o.append("  {" + ", ".join(str(int(x)) for x in mv[k]) + "}, \\")
    o.append("}")
    return "\n".join(o) + "\n"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--write-golden", action="store_true", help="rewrite tests/golden/tables.json")
    args = ap.parse_args()
    lb, gm, sp, mv = build_all()
    text = emit(lb, gm, sp, mv)
    for rel in ("corintho_ai_amd/csrc/tables.inc", "oracle/tables.inc"):
        path = os.path.join(ROOT, rel)
        with open(path, "w") as f:
            f.write(text)
        print("wrote", rel)
    if args.write_golden:
        with open(os.path.join(ROOT, "tests/golden/tables.json"), "w") as f:
            json.dump(digests(lb, gm, sp, mv), f, indent=1)
        print("wrote tests/golden/tables.json")


if __name__ == "__main__":
    sys.exit(main())
