#!/usr/bin/env python3
"""gen_mc_table.py -- build the 256-case marching-cubes triangle table used by BOTH the HIP kernels
(lidar_transfer_amd/csrc/lt_mc_table.h), which the CPU oracle includes too (oracle/Makefile: -I../lidar_transfer_amd/csrc).  BUILD / TEST INFRASTRUCTURE.

Why generated and not copied: the reference calls scikit-image's ``marching_cubes_lewiner``
(/root/reference auxiliary/fusion_lidar.py:407); scikit-image is not part of the reference, is not importable in this
image and its look-up tables cannot be read here, so the table is derived from the published algorithm (Lorensen &
Cline 1987) with a face rule that makes the surface watertight by construction:

  * corner i of a cell sits at offset (i & 1, (i >> 1) & 1, (i >> 2) & 1) along (x, y, z) = the volume's
    (dim0, dim1, dim2); bit i of the case index is set when the corner is INSIDE (tsdf < level);
  * a lattice edge is named by its lower corner c0 and its axis a (code = c0 | a << 3): exactly the (owner voxel,
    axis) pair under which the kernels number the shared vertices;
  * on every cube face the crossing points are joined by segments that depend on the four corner signs of THAT FACE
    only: one segment for two crossings; for four crossings (two inside corners on a diagonal -- the ambiguous
    face) two segments that each cut off one INSIDE corner.  Both cells that share a face see the same four signs,
    hence the same segments: no holes, for any field;
  * the directed segments chain into closed polygons (inside region to the right, seen from outside the cell, so the
    normals point towards the positive / free-space side); each polygon is triangulated without diagonals that lie
    in a cube face (the fan from its first vertex whenever that qualifies), so that no two cells ever draw the same
    diagonal: every mesh is a consistently oriented 2-manifold.

Interior ambiguities (Lewiner's case 4/6/7/10/12/13 sub-cases, centre vertex) are not resolved differently from the
faces -- topology may differ from scikit-image's there; DESIGN.md section 7c, "parity unpinned".

    python tools/gen_mc_table.py            # writes both headers
    python tools/gen_mc_table.py --check    # exit 1 if the committed headers differ
"""
import itertools
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = [os.path.join(ROOT, "lidar_transfer_amd", "csrc", "lt_mc_table.h")]  # ONE copy: the CPU oracle includes it from there


def corner_offset(i):
    return (i & 1, (i >> 1) & 1, (i >> 2) & 1)


def edge_code(ca, cb):
    """code of the lattice edge between adjacent corners ca, cb: lower corner | axis << 3"""
    d = ca ^ cb
    assert d in (1, 2, 4)
    return min(ca, cb) | ({1: 0, 2: 1, 4: 2}[d] << 3)


def faces_ccw_from_outside():
    """six faces: four corner ids in counter-clockwise order seen from OUTSIDE the cell"""
    out = []
    for axis in range(3):
        for side in (0, 1):
            u, v = [a for a in range(3) if a != axis]
            # corners of the face in (u, v) order 00, 10, 11, 01
            quad = []
            for (du, dv) in ((0, 0), (1, 0), (1, 1), (0, 1)):
                off = [0, 0, 0]
                off[axis] = side
                off[u] = du
                off[v] = dv
                quad.append(off[0] | off[1] << 1 | off[2] << 2)
            # (u, v, axis) is right-handed when (axis, u, v) is a cyclic shift of (0, 1, 2) ...
            right_handed = (axis, u, v) in ((0, 1, 2), (1, 2, 0), (2, 0, 1))
            # ... then the quad is counter-clockwise seen from +axis; flip for the side facing -axis
            ccw_from_plus = right_handed
            if (side == 1) != ccw_from_plus:
                quad = quad[::-1]
            out.append(quad)
    return out


FACES = faces_ccw_from_outside()


def segments_of_face(quad, inside):
    """directed segments (from_edge, to_edge) on one face; inside region on the RIGHT seen from outside"""
    ins = [inside[c] for c in quad]
    n_in = sum(ins)
    segs = []
    if n_in == 0 or n_in == 4:
        return segs
    # walk the quad counter-clockwise; an edge k joins quad[k] and quad[k + 1]
    crossing = [ins[k] != ins[(k + 1) % 4] for k in range(4)]

    def e(k):
        return edge_code(quad[k], quad[(k + 1) % 4])

    if sum(crossing) == 2:
        # one run of inside corners: walking counter-clockwise we ENTER the inside run at edge k_in
        # (outside -> inside) and LEAVE it at edge k_out.  The segment runs from the entering edge to the leaving
        # edge: seen from outside the cell the inside corners are then on its right (check_orientation() below
        # verifies the consequence: normals point to the positive side).
        k_in = [k for k in range(4) if crossing[k] and not ins[k]][0]
        k_out = [k for k in range(4) if crossing[k] and ins[k]][0]
        segs.append((e(k_in), e(k_out)))
    else:
        # ambiguous face: inside corners on a diagonal; cut each inside corner off on its own
        for k in range(4):
            if ins[k]:
                # corner quad[k] is entered at edge k - 1 and left at edge k
                segs.append((e((k - 1) % 4), e(k)))
    return segs


def polygons(case):
    inside = [(case >> i) & 1 for i in range(8)]
    nxt = {}
    for quad in FACES:
        for a, b in segments_of_face(quad, inside):
            assert a not in nxt, (case, a)
            nxt[a] = b
    polys = []
    seen = set()
    for start in sorted(nxt):
        if start in seen:
            continue
        loop = [start]
        seen.add(start)
        cur = nxt[start]
        while cur != start:
            assert cur not in seen
            loop.append(cur)
            seen.add(cur)
            cur = nxt[cur]
        polys.append(loop)
    # every crossing edge is used exactly once
    crossing_edges = {edge_code(a, b) for a in range(8) for b in range(8)
                      if a < b and (a ^ b) in (1, 2, 4) and inside[a] != inside[b]}
    assert seen == crossing_edges, case
    return polys


def edge_mid(code):
    c0, axis = code & 7, code >> 3
    p = list(corner_offset(c0))
    p[axis] += 0.5
    return p


def edge_faces(code):
    """the two cube faces (axis, side) a lattice edge lies on"""
    c0, axis = code & 7, code >> 3
    return {(u, (c0 >> u) & 1) for u in range(3) if u != axis}


def triangulations(n):
    """all triangulations of a convex n-gon as lists of index triples (i < j < k), in a fixed order that
    starts with the fan from vertex 0"""
    def rec(lo, hi):  # polygon lo, lo+1, ..., hi (chord lo-hi closes it)
        if hi - lo < 2:
            yield []
            return
        for mid in range(hi - 1, lo, -1):
            for left in rec(lo, mid):
                for right in rec(mid, hi):
                    yield left + [(lo, mid, hi)] + right
    return list(rec(0, n - 1))


def triangulate(loop):
    """A diagonal joining two vertices that lie on the same cube face would lie IN that face, where the neighbouring
    cell may draw the very same diagonal (ambiguous faces): two sheets sharing an edge.  Take the first triangulation
    without such a diagonal (the fan from vertex 0 when it qualifies)."""
    n = len(loop)
    best, best_bad = None, None
    for tri in triangulations(n):
        bad = 0
        for (i, j, k) in tri:
            for a, b in ((i, j), (j, k), (i, k)):
                if (b - a) % n in (1, n - 1):
                    continue  # polygon side, not a diagonal
                if edge_faces(loop[a]) & edge_faces(loop[b]):
                    bad += 1
        if best is None or bad < best_bad:
            best, best_bad = tri, bad
        if bad == 0:
            break
    return [(loop[i], loop[j], loop[k]) for (i, j, k) in best], best_bad


def triangulate_other(loop):
    """The triangulation that shares the FEWEST diagonals with triangulate()'s choice (a quad: its other diagonal; a
    pentagon: a fan from another vertex, no diagonal in common; ...), whether or not a diagonal lies in a cube face.  The
    variant table built from it bounds what ANY other choice of diagonals -- scikit-image's Lewiner table included -- can
    change on the cases whose polygons are the same (tests/test_mc_gpu.py)."""
    n = len(loop)
    if n < 4:
        return triangulate(loop)[0]
    base, _ = triangulate(loop)
    pos = {code: i for i, code in enumerate(loop)}

    def diagonals(tri_idx):
        d = set()
        for (i, j, k) in tri_idx:
            for a, b in ((i, j), (j, k), (i, k)):
                if (b - a) % n not in (1, n - 1):
                    d.add((min(a, b), max(a, b)))
        return d
    base_d = diagonals([tuple(pos[c] for c in t) for t in base])
    best, best_shared = None, None
    for tri in triangulations(n):
        shared = len(diagonals(tri) & base_d)
        if best is None or shared < best_shared:
            best, best_shared = tri, shared
    return [(loop[i], loop[j], loop[k]) for (i, j, k) in best]


def triangles(case, variant="default"):
    tris = []
    for loop in polygons(case):
        tris.extend(triangulate(loop)[0] if variant == "default" else triangulate_other(loop))
    return tris


def packed_words(variant="default"):
    """The table in LT_MC_PACKED's layout (two 64-bit words per case) as a list of 512 ints."""
    out = []
    for c in range(256):
        rows = triangles(c, variant)
        v = len(rows)
        for i, code in enumerate(x for t in rows for x in t):
            v |= code << (8 + 5 * i)
        out += [v & ((1 << 64) - 1), v >> 64]
    return out


def case_classes():
    """Per case: is it where a Lewiner table (the reference's scikit-image call, fusion_lidar.py:407) CAN differ in
    topology -- a face with its two inside corners on a diagonal (face ambiguity: Lewiner's cases 3, 6, 7, 10, 12, 13) or two
    inside / outside corners on a space diagonal and nothing else (interior ambiguity alone: case 4) -- or only in the choice
    of diagonals (a polygon with more than three vertices), or not at all (triangles only)."""
    out = []
    for c in range(256):
        inside = [(c >> i) & 1 for i in range(8)]
        face_amb = False
        for quad in FACES:
            ins = [inside[q] for q in quad]
            if sum(ins) == 2 and ins[0] == ins[2]:
                face_amb = True
        n_in = sum(inside)
        diag4 = False
        if n_in in (2, 6):
            minority = [i for i in range(8) if inside[i] == (1 if n_in == 2 else 0)]
            diag4 = (minority[0] ^ minority[1]) == 7
        polys = polygons(c) if 0 < c < 255 else []
        out.append({"face_ambiguous": face_amb, "interior_ambiguous_only": diag4 and not face_amb,
                    "splits_a_polygon": any(len(p) > 3 for p in polys), "n_polygons": len(polys),
                    "n_triangles": sum(len(p) - 2 for p in polys)})
    return out


def check_orientation():
    """single inside corner 0: the triangle's normal must point away from it (towards the positive side)"""
    (a, b, c), = triangles(1)
    pa, pb, pc = edge_mid(a), edge_mid(b), edge_mid(c)
    u = [pb[i] - pa[i] for i in range(3)]
    v = [pc[i] - pa[i] for i in range(3)]
    n = [u[1] * v[2] - u[2] * v[1], u[2] * v[0] - u[0] * v[2], u[0] * v[1] - u[1] * v[0]]
    return sum(n) > 0  # corner 0 is the origin: away from it = positive components


def render():
    assert check_orientation(), "orientation convention broken"
    rows = [triangles(c) for c in range(256)]
    assert all(len(triangles(c, "other")) == len(rows[c]) for c in range(256))
    max_t = max(len(r) for r in rows)
    lines = []
    lines.append("/* GENERATED by tools/gen_mc_table.py -- do not edit.  256-case marching-cubes table, face-consistent")
    lines.append(" * (watertight) disambiguation; see the generator for the conventions:")
    lines.append(" *   corner i at offset (i & 1, (i >> 1) & 1, (i >> 2) & 1) along (x, y, z); case bit i = corner i inside (< level);")
    lines.append(" *   a triangle vertex is a lattice edge code = lower corner | axis << 3 (owner voxel offset, edge axis);")
    lines.append(" *   normals point to the positive side. */")
    lines.append("#ifndef LT_MC_TABLE_H")
    lines.append("#define LT_MC_TABLE_H")
    lines.append("#ifndef LT_TABLE_ATTR")
    lines.append("#define LT_TABLE_ATTR /* e.g. __device__ when included from HIP code */")
    lines.append("#endif")
    lines.append(f"#define LT_MC_MAX_TRIS {max_t}")
    lines.append("LT_TABLE_ATTR static const unsigned char LT_MC_NTRIS[256] = {")
    for r0 in range(0, 256, 32):
        lines.append(" ".join(f"{len(rows[c])}," for c in range(r0, r0 + 32)))
    lines.append("};")
    lines.append("/* first entry of case c in LT_MC_TRIS (3 edge codes per triangle) */")
    lines.append("LT_TABLE_ATTR static const unsigned short LT_MC_FIRST[257] = {")
    first = [0]
    for r in rows:
        first.append(first[-1] + 3 * len(r))
    for r0 in range(0, 257, 16):
        lines.append(" ".join(f"{x}," for x in first[r0:r0 + 16]))
    lines.append("};")
    lines.append(f"LT_TABLE_ATTR static const unsigned char LT_MC_TRIS[{max(first[-1], 1)}] = {{")
    flat = [code for r in rows for t in r for code in t]
    for r0 in range(0, len(flat), 24):
        lines.append(" ".join(f"{x}," for x in flat[r0:r0 + 24]))
    lines.append("};")
    lines.append("/* the same, one 128-bit word per case: bits 0..2 = number of triangles, bits 8 + 5 i .. 12 + 5 i = edge code i")
    lines.append(" * (i = 3 * triangle + corner): one load hands a lane its whole cell */")
    lines.append("LT_TABLE_ATTR static const unsigned long long LT_MC_PACKED[512] = {")
    for c0 in range(0, 256, 2):
        parts = []
        for c in (c0, c0 + 1):
            v = len(rows[c])
            for i, code in enumerate(x for t in rows[c] for x in t):
                v |= code << (8 + 5 * i)
            parts.append("0x%016xull, 0x%016xull," % (v & ((1 << 64) - 1), v >> 64))
        lines.append(" ".join(parts))
    lines.append("};")
    lines.append("#endif")
    return "\n".join(lines) + "\n"


if __name__ == "__main__":
    text = render()
    if "--check" in sys.argv:
        bad = [p for p in OUT if not os.path.exists(p) or open(p).read() != text]
        if bad:
            print("stale:", bad)
            sys.exit(1)
        print("lt_mc_table.h up to date")
    else:
        for p in OUT:
            with open(p, "w") as f:
                f.write(text)
        rows = [triangles(c) for c in range(256)]
        in_face = sum(triangulate(loop)[1] for c in range(256) for loop in polygons(c))
        print("diagonals lying in a cube face (all cases):", in_face)
        print("max triangles per cell:", max(len(r) for r in rows), "total entries:", sum(3 * len(r) for r in rows))
