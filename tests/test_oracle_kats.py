"""Pin the CPU oracle against the reference's own known-answer tests.

Each test cites the reference test it restates (paths relative to /root/reference).
"""
import numpy as np
import pytest

import oracle_lib as O
from point_cloud_viewer_amd import synthetic

U8, U16, F32, F64 = 1, 2, 3, 4


def _roundtrip(enc, value, mn, edge):
    L = O.lib()
    return L.pcvo_decode_coord(enc, L.pcvo_encode_coord(enc, value, mn, edge), mn, edge)


def test_codec_plain_scalar():
    # src/read_write/codec.rs:158-181 plain_scalar
    value, mn, edge = 41.33333, 40.0, 2.0
    assert abs(_roundtrip(F32, value, mn, edge) - value) < 1e-7
    assert abs(_roundtrip(F64, value, mn, edge) - value) < 1e-14


def test_codec_fixpoint_scalar():
    # src/read_write/codec.rs:183-212 fixpoint_scalar (u8 / u16 arms; u32 is not an on-disk encoding)
    value, mn, edge = 41.33333, 40.0, 2.0
    assert abs(_roundtrip(U8, value, mn, edge) - value) < 1e-2
    assert abs(_roundtrip(U16, value, mn, edge) - value) < 1e-4


def test_codec_truncation_rule():
    # simba 0.2.1 try_convert == `as` cast: truncation toward zero, saturation, NaN -> 0
    L = O.lib()
    assert L.pcvo_encode_coord(U8, 0.999999, 0.0, 1.0) == 254  # 254.9997 truncates
    assert L.pcvo_encode_coord(U8, 1.0, 0.0, 1.0) == 255
    assert L.pcvo_encode_coord(U8, 7.0, 0.0, 1.0) == 255  # clamp
    assert L.pcvo_encode_coord(U8, -3.0, 0.0, 1.0) == 0
    assert L.pcvo_encode_coord(U8, float("nan"), 0.0, 1.0) == 0
    assert L.pcvo_encode_coord(U16, 0.5, 0.0, 1.0) == 32767  # 32767.5 truncates


def test_position_encoding_thresholds():
    # src/read_write/codec.rs:31-40
    L = O.lib()
    assert L.pcvo_position_encoding(0.2, 0.001) == U8  # log2(200)=7.6 -> 7+1 = 8 bits
    assert L.pcvo_position_encoding(0.3, 0.001) == U16  # log2(300)=8.2 -> 9 bits
    assert L.pcvo_position_encoding(60.0, 0.001) == U16  # 15.87 -> 16
    assert L.pcvo_position_encoding(70.0, 0.001) == F32  # 16.09 -> 17
    assert L.pcvo_position_encoding(16000.0, 0.001) == F32  # 23.9 -> 24
    assert L.pcvo_position_encoding(17000.0, 0.001) == F64  # 24.02 -> 25
    assert L.pcvo_position_encoding(0.0005, 0.001) == U8  # negative log2 saturates to 0 -> 1 bit


def test_parent_node_name():
    # src/octree/node.rs:278-283
    hi, lo = O.node_id_from_str("r123456")
    import ctypes as C
    phi, plo = C.c_uint64(), C.c_uint64()
    assert O.lib().pcvo_node_id_parent(hi, lo, C.byref(phi), C.byref(plo)) == 1
    assert (phi.value, plo.value) == O.node_id_from_str("r12345")
    assert O.node_id_str(phi.value, plo.value) == "r12345"


def test_child_index():
    # src/octree/node.rs:285-296
    import ctypes as C
    L = O.lib()
    assert L.pcvo_node_id_child_index(*O.node_id_from_str("r123451")) == 1
    assert L.pcvo_node_id_child_index(*O.node_id_from_str("r123457")) == 7
    phi, plo = C.c_uint64(), C.c_uint64()
    assert L.pcvo_node_id_parent(*O.node_id_from_str("r"), C.byref(phi), C.byref(plo)) == 0  # root: None


def test_bounding_box_of_node_ids():
    # src/octree/node.rs:298-317 — defines x = bit 2, y = bit 1, z = bit 0
    import ctypes as C
    L = O.lib()
    root_min = np.array([-5.0, -5.0, -5.0])
    out = np.zeros(3)
    edge = C.c_double()
    L.pcvo_find_bounding_cube(*O.node_id_from_str("r0"), O._d(root_min), 10.0, O._d(out), C.byref(edge))
    assert out.tolist() == [-5.0, -5.0, -5.0] and edge.value == 5.0
    L.pcvo_find_bounding_cube(*O.node_id_from_str("r13"), O._d(root_min), 10.0, O._d(out), C.byref(edge))
    assert out.tolist() == [-5.0, -2.5, 2.5] and edge.value == 2.5


def test_node_id_layout():
    # src/octree/node.rs:101-111: u128 = level << 120 | index ; proto high/low halves
    hi, lo = O.node_id_from_str("r13")
    assert hi == (2 << 56) and lo == 0o13
    assert O.node_id_str(hi, lo) == "r13"
    deep = "r" + "7" * 30
    hi, lo = O.node_id_from_str(deep)
    assert hi >> 56 == 30 and O.node_id_str(hi, lo) == deep


@pytest.mark.parametrize("mode", ["literal", "closed"])
def test_reference_octree_unit_test_cloud(mode):
    # src/octree/tests.rs:18-46 build_test_octree; expected answer derived in SURVEY.md §8c (5):
    # root cube min (-200,-40,0) edge 200, all levels Uint8; r=12 501, r0=0 points (no files), r4=87 500.
    x, y, z, rgb, bmin, bmax, res = synthetic.reference_unit_test_cloud()
    build = O.build_literal if mode == "literal" else O.build_closed
    t = build(res, bmin, bmax, x, y, z, rgb)
    assert t.version == 13
    assert t.total_points() == 100001  # tests.rs:100-101 expects every point back
    assert sorted(t.nodes) == ["r", "r0", "r4"]
    assert t.nodes["r"]["num_points"] == 12501
    assert t.nodes["r0"]["num_points"] == 0 and t.nodes["r0"]["files"] == 0
    assert t.nodes["r4"]["num_points"] == 87500
    assert all(n["encoding"] == U8 for n in t.nodes.values())
    assert len(t.nodes["r4"]["xyz"]) == 87500 * 3 and len(t.nodes["r4"]["rgb"]) == 87500 * 3
    # colours are passed through untouched
    assert set(t.nodes["r"]["rgb"][0::3]) == {255} and set(t.nodes["r"]["rgb"][1::3]) == {0}


def test_sum_of_num_points_equals_input():
    # point_cloud_test/tests/main.rs:10-23 num_points_in_octree_meta (at reduced N)
    n = 200_000
    x, y, z, rgb, bmin, bmax = synthetic.uniform_ecef(n)
    t = O.build_closed(0.001, bmin, bmax, x, y, z, rgb, threads=4)
    assert t.total_points() == n


def test_double_double_reciprocal_division_is_exact_for_every_code():
    """The HIP kernels decode with q = fma(v, yh, v * yl), (yh, yl) the double-double reciprocal of max, instead of an
    IEEE division (pcv_chain_dev.h pcv_div_code, PCV_RECIP_255 / PCV_RECIP_65535). Exhaustive proof for all u8 / u16
    codes, FMA emulated exactly; also pins the two constants."""
    from fractions import Fraction as F

    def fma(a, b, c):
        e = F(a) * F(b) + F(c)
        return float(e) if e != 0 else 0.0

    for m, want in ((255.0, (float.fromhex("0x1.0101010101010p-8"), float.fromhex("0x1.0101010101010p-64"))),
                    (65535.0, (float.fromhex("0x1.0001000100010p-16"), float.fromhex("0x1.0001000100010p-80")))):
        yh = 1.0 / m
        yl = fma(-m, yh, 1.0) / m
        assert (yh, yl) == want
        for v in range(int(m) + 1):
            v = float(v)
            assert fma(v, yh, v * yl) == v / m, (m, v)


def test_octant_digit_from_integer_codes_at_the_admitted_ratio():
    """csrc/pcv_chain_dev.h pcv_digit_from_codes: for a u8 / u16-coded level the next octant digit (node.rs:34-42,
    strict > against aabb.rs:175-192's centre) of the decoded position fma(RN(c / M), e, mn) (codec.rs:124-139) equals
    c > M // 2 wherever pcv_make_levels admits the shortcut: (2.5 A / e + 3) * 4.04 * M < 2^53. Replayed in exact
    rational arithmetic (Fraction -> float is correctly rounded, i.e. the FMA) for the two codes next to the centre and
    random ones, with |mn| / e right at the admitted bound (and at easy ratios)."""
    from fractions import Fraction
    import math
    import random
    rnd = random.Random(7)
    for M, half in ((255, 127), (65535, 32767)):
        limit = (2.0 ** 53 / (4.04 * M) - 3.0) / 2.5  # largest admitted A / e
        for trial in range(4000):
            e = math.ldexp(rnd.uniform(1.0, 2.0), rnd.randint(-20, 12))
            ratio = limit * (1.0 - 1e-9) if trial % 2 == 0 else rnd.uniform(0.0, limit)
            # cube min with |mn| + e <= A = ratio * e, either sign, arbitrary low bits
            mag = max(0.0, ratio * e - e) * (1.0 if trial % 4 < 2 else rnd.uniform(0.0, 1.0))
            mn = math.copysign(mag, rnd.choice((-1.0, 1.0)))
            if mn < 0:
                mn = -(mag + e) if mag + e <= ratio * e else mn  # keep |mn + e| inside A as well
            centre = (mn + (mn + e)) / 2.0
            for c in (half, half + 1, 0, M, rnd.randint(0, M), rnd.randint(0, M)):
                q = c / M  # IEEE division == pcv_div_code (test_exact_code_division above)
                p = float(Fraction(mn) + Fraction(q) * Fraction(e))  # one rounding: the FMA
                assert (p > centre) == (c > half), (M, c, mn, e, p, centre)


def test_octant_digit_from_float32_codes_at_the_admitted_ratio():
    """csrc/pcv_chain_dev.h pcv_bits_from_codes / pcv_f32_code_tie (round 4): for a Float32-coded level the next octant
    digit (node.rs:34-42, strict > against aabb.rs:175-192's centre) of the decoded position fma((double)(float)t, e, mn)
    (codec.rs:115-121, 131-135) equals v > 0.5 for every code v != 0.5 wherever pcv_make_levels admits the shortcut:
    (2.5 A / e + 3) * 4.04 * 2^24 < 2^53 (PcvLevels::digit_mode == 2). A code of exactly 0.5 is a tie of the exact values —
    there the rounded comparison decides, which is why the kernel falls back to it; the test shows that both outcomes
    occur. Replayed in exact rational arithmetic for the floats next to 0.5 and random ones, |mn| / e at the admitted bound."""
    from fractions import Fraction
    import math
    import random
    import numpy as np
    rnd = random.Random(11)
    limit = (2.0 ** 53 / (4.04 * 2.0 ** 24) - 3.0) / 2.5  # largest admitted A / e
    below, above = float(np.nextafter(np.float32(0.5), np.float32(0.0))), float(np.nextafter(np.float32(0.5), np.float32(1.0)))
    assert below == 0.5 - 2.0 ** -25 and above == 0.5 + 2.0 ** -24
    tie_outcomes = set()
    for trial in range(6000):
        e = math.ldexp(rnd.uniform(1.0, 2.0), rnd.randint(-10, 14))
        ratio = limit * (1.0 - 1e-9) if trial % 2 == 0 else rnd.uniform(0.0, limit)
        mag = max(0.0, ratio * e - e) * (1.0 if trial % 4 < 2 else rnd.uniform(0.0, 1.0))
        mn = math.copysign(mag, rnd.choice((-1.0, 1.0)))
        if mn < 0:
            mn = -(mag + e) if mag + e <= ratio * e else mn  # keep |mn + e| inside A as well
        centre = (mn + (mn + e)) / 2.0
        codes = [below, above, 0.0, 1.0, float(np.float32(rnd.random())), float(np.float32(rnd.uniform(0.49999, 0.50001)))]
        for v in codes:
            p = float(Fraction(mn) + Fraction(v) * Fraction(e))  # one rounding: the FMA of the decode
            if v == 0.5:
                continue
            assert (p > centre) == (v > 0.5), (v, mn, e, p, centre)
        p = float(Fraction(mn) + Fraction(0.5) * Fraction(e))
        tie_outcomes.add(p > centre)
    assert tie_outcomes == {False, True} or tie_outcomes == {False}, tie_outcomes  # the tie is decided by rounding, not by the code


def test_float32_codes_from_codes_at_the_admitted_threshold():
    """csrc/pcv_encode.hip CP_CODE_LOOP / pcv_make_levels (round 5): where the table admits the step, the Float32 code of
    level k + 1 — (float)clamp((p_k - min_{k+1}) / edge_{k+1}, 0, 1) with p_k = fma(v, edge_k, min_k) the decoded level-k
    position (codec.rs:115-121, 131-135; node.rs:157-172 for the cube) — equals w = 2 v - bit for every level-k code v whose
    w is at least the level's threshold and below 1 (bit = v > 1/2). The thresholds are the LIBRARY's (pcv_level_shortcuts),
    the chain is replayed in exact rational arithmetic (Fraction -> float is correctly rounded: the FMA, the subtraction
    and the IEEE division), for cubes at the magnitudes the table was built for: the bench cloud's and ECEF's, with the
    cube min at the far end of the root cube (largest rounding errors), w at and next to the threshold, next to 1 and random.
    Below the threshold the prediction does fail (the reason the kernel runs the full step there): the test finds such a case."""
    from fractions import Fraction as F
    import math
    import random
    import point_cloud_viewer_amd as pcv
    rnd = random.Random(5)
    f32 = np.float32
    failures_below = 0
    admitted_steps = 0
    checked = 0
    for bmin, extent, res in (((3.0, -5.0, 1.0), 1093.7, 0.001),            # the bench cloud's magnitudes
                              ((-2.7e6, -4.3e6, 3.8e6), 1100.0, 0.001),   # ECEF
                              ((4.1e6, 4.8e6, 5.0e5), 8100.0, 0.001),     # ECEF, bigger cube
                              ((-50.0, -50.0, -50.0), 100.0, 1e-4)):
        bmax = tuple(b + extent for b in bmin)
        ml, edge, enc = pcv.level_table(bmin, bmax, res, cap=40)
        _, mode, thr = pcv.level_shortcuts(bmin, bmax, res)
        for k in range(1, min(ml, 21)):
            if thr[k] == 0.0:
                continue
            admitted_steps += 1
            assert enc[k] == 3 and enc[k + 1] == 3 and mode[k] == 2, (k, enc[k], enc[k + 1], mode[k])  # PCV_ENC_FLOAT32
            t = float(thr[k])
            assert math.frexp(t)[0] == 0.5 and t <= 2.0 ** -8  # a power of two
            e, e2 = float(edge[k]), float(edge[k + 1])
            assert e2 * 2.0 == e
            for trial in range(300):
                # a level-k cube min somewhere in the root cube, biased to the corner with the largest magnitudes, arbitrary low bits
                a = rnd.randrange(3)
                lo, hi = bmin[a], bmin[a] + extent - e
                m = (hi if abs(hi) > abs(lo) else lo) if trial % 2 == 0 else rnd.uniform(lo, hi)
                m = float(np.nextafter(m, rnd.choice((-math.inf, math.inf)))) if trial % 3 == 0 else m
                up = lambda f: float(np.nextafter(f32(f), f32(2.0)))
                half_t = 0.5 + t / 2.0
                hv = float(f32(half_t)) if float(f32(half_t)) >= half_t else up(half_t)  # smallest float32 v > 1/2 with 2 v - 1 >= t
                vs = [t / 2.0, up(t / 2.0), float(f32(rnd.uniform(t / 2.0, 0.5))), float(f32(rnd.uniform(t / 2.0, 2.0 * t))),
                      float(np.nextafter(f32(0.5), f32(0.0))),  # w next to 1 from below (bit 0)
                      hv, up(hv), float(f32(rnd.uniform(hv, 1.0))), float(np.nextafter(f32(1.0), f32(0.0))), float(f32(rnd.random()))]
                for v in vs:
                    assert float(f32(v)) == v
                    bit = 1 if v > 0.5 else 0
                    w = 2.0 * v - bit  # exact
                    if not (t <= w < 1.0):
                        continue  # the kernel runs the full step for this wave
                    assert float(f32(w)) == w
                    p = float(F(m) + F(v) * F(e))            # decode: one rounding
                    m2 = float(F(m) + bit * F(e2))           # pcv_step_min: one rounding
                    x = float(F(p) - F(m2))
                    q = float(F(x) / F(e2)) if x != 0 else 0.0
                    got = float(f32(min(max(q, 0.0), 1.0)))
                    assert got == w, (k, m, e, v, w, q, got)
                    checked += 1
                # below the threshold the chain's rounding noise does show: count the misses (never asserted to be zero)
                w = float(f32(rnd.uniform(0.0, t * 2.0 ** -10)))
                if w > 0.0:
                    v = w / 2.0
                    p = float(F(m) + F(v) * F(e))
                    x = float(F(p) - F(m))
                    q = float(F(x) / F(e2)) if x != 0 else 0.0
                    failures_below += float(f32(min(max(q, 0.0), 1.0))) != w
    assert admitted_steps >= 8 and checked > 20000, (admitted_steps, checked)
    assert failures_below > 0  # the threshold is not vacuous
