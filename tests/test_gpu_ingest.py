"""Streaming batch ingest at the C ABI: batches in the reference's own layout — `PointsBatch`
with `position: Vec<Point3<f64>>` AoS, colour Vec<Vector3<u8>>, intensity Vec<f32> (src/lib.rs:102-107), 500 000 points
at a time (src/lib.rs:52), fed to `build_octree(.., input: impl Iterator<Item = PointsBatch>, ..)` (generation.rs:289-295).
The streamed build must equal the one-shot build and the oracle byte for byte, whatever the batch sizes."""
import os

import numpy as np
import pytest

import oracle_lib as O
import point_cloud_viewer_amd as pcv
from point_cloud_viewer_amd import synthetic

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = pcv.Context(0)
    yield c
    c.close()


def batches(x, y, z, rgb, inten, sizes):
    """The cloud cut into PointsBatch-shaped pieces: position (n, 3) AoS."""
    pos = np.stack([x, y, z], axis=1)
    at = 0
    for s in sizes:
        yield dict(position=pos[at:at + s], color=rgb[at:at + s], intensity=None if inten is None else inten[at:at + s])
        at += s
    assert at == x.size


def cut(n, batch):
    sizes = [batch] * (n // batch)
    if n % batch:
        sizes.append(n % batch)  # the ragged last batch
    return sizes


def same_tree(got, want, check_intensity=False):
    assert set(got) == set(want), sorted(set(got) ^ set(want))[:10]
    for name, nd in want.items():
        g = got[name]
        for k in ("num_points", "encoding", "xyz", "rgb") + (("intensity",) if check_intensity else ()):
            assert g[k] == nd[k], (name, k)


def test_batches_of_500k_with_a_ragged_last_batch_equal_the_one_shot_build_and_the_oracle(ctx):
    n = 2_300_017  # 4 batches of 500 000 (src/lib.rs:52) + a ragged one of 300 017
    x, y, z, rgb, bmin, bmax = synthetic.gaussian_clusters(n, seed=61, num_clusters=7, extent=400.0, sigma_range=(0.2, 9.0))
    cap = 20_000
    with O.max_points_per_node(cap):
        want = O.build_closed(0.001, bmin, bmax, x, y, z, rgb, threads=8).nodes
    one_shot = ctx.build(0.001, pcv.Aabb(bmin, bmax), x, y, z, rgb, max_points_per_node=cap).to_dict()
    same_tree(one_shot, want)
    ing = ctx.ingest(n, has_intensity=False)
    for b in batches(x, y, z, rgb, None, cut(n, 500_000)):
        ing.append(b["position"], b["color"])
    assert ing.num_points == n
    lo, hi = ing.bbox()  # find_bounding_box folded batch by batch == K1 over the whole cloud == numpy
    assert np.array_equal(lo, [x.min(), y.min(), z.min()]) and np.array_equal(hi, [x.max(), y.max(), z.max()])
    k1 = ctx.aabb_reduce(x, y, z)
    assert np.array_equal(lo, k1[0]) and np.array_equal(hi, k1[1])
    tree = ing.finish(0.001, pcv.Aabb(bmin, bmax), max_points_per_node=cap)
    same_tree(tree.to_dict(), want)
    assert tree.num_points == n
    tree.free()


@pytest.mark.parametrize("sizes", [[1], [1, 1, 1, 7, 64, 1023, 1024, 1025], [3, 700_001, 2, 299_994],
                                   [1_300_000, 5], [0, 17, 0, 0, 50_000, 0]])
def test_any_batch_sizes_odd_offsets_and_the_box_from_the_ingest(ctx, sizes):
    """Odd batch sizes put every later batch at odd offsets of the device arrays (colour at byte offsets of any alignment);
    batches above 2^20 points are split inside; empty batches are no-ops; bounding_box=None takes the box of the ingest."""
    n = sum(sizes)
    x, y, z, rgb, bmin, bmax = synthetic.gaussian_clusters(max(n, 2), seed=62 + len(sizes), num_clusters=3, extent=120.0,
                                                           sigma_range=(0.1, 4.0), offset=(-2.7e6, -4.3e6, 3.8e6))
    x, y, z, rgb = x[:n], y[:n], z[:n], rgb[:n]
    inten = (np.arange(n, dtype=np.int64) * 7919 % 10_007).astype(np.float32) * 0.125 - 3.0
    ing = ctx.ingest(0 if len(sizes) % 2 else n, has_intensity=True)  # with and without the NumberOfPoints hint
    for b in batches(x, y, z, rgb, inten, sizes):
        ing.append(b["position"], b["color"], b["intensity"])
    assert ing.num_points == n
    tree = ing.finish(0.001, None, max_points_per_node=3_000)
    meta = tree.meta()
    tight_min, tight_max = np.array([x.min(), y.min(), z.min()]), np.array([x.max(), y.max(), z.max()])
    assert np.array_equal(meta["bbox_min"], tight_min) and np.array_equal(meta["bbox_max"], tight_max)
    with O.max_points_per_node(3_000):
        want = O.build_closed(0.001, tight_min, tight_max, x, y, z, rgb, inten, threads=8).nodes
    same_tree(tree.to_dict(), want, check_intensity=True)
    tree.free()


def test_the_stream_may_be_longer_than_the_hint(ctx):
    """NumberOfPoints is a hint: the device arrays grow (device-to-device move) and nothing is lost."""
    n = 2_600_000
    x, y, z, rgb, bmin, bmax = synthetic.gaussian_clusters(n, seed=70, num_clusters=4, extent=300.0, sigma_range=(0.3, 6.0))
    ing = ctx.ingest(100_000, has_intensity=False)  # forces two growth steps (1 M floor, then x 1.5, then the need)
    for b in batches(x, y, z, rgb, None, cut(n, 450_000)):
        ing.append(b["position"], b["color"])
    tree = ing.finish(0.001, pcv.Aabb(bmin, bmax), max_points_per_node=25_000)
    got = tree.to_dict()
    ref = ctx.build(0.001, pcv.Aabb(bmin, bmax), x, y, z, rgb, max_points_per_node=25_000).to_dict()
    same_tree(got, ref)
    tree.free()


def test_other_calls_between_appends_do_not_touch_the_batches_in_hand(ctx):
    """Batches are packed into a pinned chunk and go up when it is full: a build from host arrays (which takes chunks of the same
    ring), a second ingest and a bbox query in the middle of the stream must leave the batches still in hand alone."""
    n = 900_000
    x, y, z, rgb, bmin, bmax = synthetic.gaussian_clusters(n, seed=74, num_clusters=5, extent=150.0, sigma_range=(0.2, 5.0))
    ref = ctx.build(0.001, pcv.Aabb(bmin, bmax), x, y, z, rgb, max_points_per_node=15_000).to_dict()
    pos = np.stack([x, y, z], axis=1)
    a = ctx.ingest(n, has_intensity=False)
    b = ctx.ingest(0, has_intensity=False)
    cuts = [0, 100_000, 100_007, 400_000, 650_000, n]
    for k in range(len(cuts) - 1):
        lo, hi = cuts[k], cuts[k + 1]
        a.append(pos[lo:hi], rgb[lo:hi])  # stays in the chunk in hand (13 MB at most here)
        if k == 0:
            other = ctx.build(0.001, pcv.Aabb(bmin, bmax), x[:300_000], y[:300_000], z[:300_000], rgb[:300_000])  # host arrays: the ring
            assert other.num_points == 300_000
            other.free()
        if k == 1:
            blo, bhi = a.bbox()  # flushes what is in hand
            assert np.array_equal(blo, [x[:hi].min(), y[:hi].min(), z[:hi].min()]) and np.array_equal(bhi, [x[:hi].max(), y[:hi].max(), z[:hi].max()])
        b.append(pos[lo:hi], rgb[lo:hi])  # a second ingest of the same context, interleaved
    ta = a.finish(0.001, pcv.Aabb(bmin, bmax), max_points_per_node=15_000)
    tb = b.finish(0.001, pcv.Aabb(bmin, bmax), max_points_per_node=15_000)
    same_tree(ta.to_dict(), ref)
    same_tree(tb.to_dict(), ref)
    ta.free()
    tb.free()


def test_build_octree_from_an_iterator_of_batches_writes_the_reference_directory(ctx, tmp_path):
    """The Python mirror of build_octree(dir, resolution, bbox, impl Iterator<Item = PointsBatch>, attributes):
    the directory equals the oracle's literal file-streaming build of the same cloud."""
    n = 700_003
    x, y, z, rgb, bmin, bmax = synthetic.gaussian_clusters(n, seed=71, num_clusters=5, extent=200.0, sigma_range=(0.1, 5.0))
    inten = np.linspace(-1.0, 250.0, n).astype(np.float32)

    class Stream:  # Iterator<Item = PointsBatch> + NumberOfPoints
        def num_points(self):
            return n

        def __iter__(self):
            return batches(x, y, z, rgb, inten, cut(n, 100_000))

    out = tmp_path / "octree"
    tree = pcv.build_octree(str(out), 0.001, pcv.Aabb(bmin, bmax), Stream(), attributes=("color", "intensity"), ctx=ctx)
    tree.free()
    want = tmp_path / "want"
    O.build_literal_dir(str(want), 0.001, bmin, bmax, x, y, z, rgb, inten, threads=8)
    names = sorted(os.listdir(want))
    assert sorted(os.listdir(out)) == names and len(names) > 20
    for name in names:
        if name != "meta.pb":  # the reference writes the node list in a nondeterministic order
            assert (out / name).read_bytes() == (want / name).read_bytes(), name


def test_misuse_is_reported_not_crashed(ctx):
    ing = ctx.ingest(10, has_intensity=True)
    pos, col = np.zeros((4, 3)), np.zeros((4, 3), np.uint8)
    with pytest.raises(ValueError):
        ing.append(pos, col)  # intensity promised, not delivered
    with pytest.raises(ValueError):
        ing.append(np.zeros((4, 2)), col, np.zeros(4, np.float32))
    ing.abort()
    ing = ctx.ingest(0, has_intensity=False)
    tree = ing.finish(0.001, None)  # an empty stream: the reference writes a root-only octree (generation.rs:312-323)
    assert tree.num_points == 0
    tree.free()
    with pytest.raises(ValueError):
        ing.append(pos, col)  # consumed
    # the context is still good
    x, y, z, rgb, bmin, bmax = synthetic.gaussian_clusters(50_000, seed=72, num_clusters=2, extent=30.0, sigma_range=(0.1, 1.0))
    ctx.build(0.001, pcv.Aabb(bmin, bmax), x, y, z, rgb).free()


def test_c_host_binary_streams_batches_to_the_same_directory(tmp_path):
    """examples/ingest_batches.c (gcc -std=c11): the library entry build_octree(dir, resolution, bbox, batches, attributes)
    over the C ABI in a separate non-Python process, one 100 000-point AoS batch of host memory at a time — the directory
    equals the oracle's literal build with the bounding box of the cloud."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "bin", "ingest_batches")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(root, "examples")])
    n = 350_017
    x, y, z, rgb, _, _ = synthetic.gaussian_clusters(n, seed=73, num_clusters=4, extent=80.0, sigma_range=(0.05, 3.0))
    inten = (np.arange(n) % 29).astype(np.float32) * 0.5
    np.stack([x, y, z], axis=1).astype("<f8").tofile(tmp_path / "xyz.f64")
    rgb.tofile(tmp_path / "rgb.u8")
    inten.astype("<f4").tofile(tmp_path / "int.f32")
    p = subprocess.run([exe, str(tmp_path / "xyz.f64"), str(tmp_path / "rgb.u8"), str(tmp_path / "int.f32"), str(n), "100000",
                        str(tmp_path / "c_out"), "0.001"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    assert f"{n} points in batches of 100000" in p.stdout
    bmin, bmax = O.aabb(x, y, z)
    O.build_literal_dir(tmp_path / "cpu", 0.001, bmin, bmax, x, y, z, rgb, inten, threads=4)
    assert not O.compare_octrees(O.load_dir(tmp_path / "c_out"), O.load_dir(tmp_path / "cpu"))
