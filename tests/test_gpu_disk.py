"""The octree that lives on disk: point queries on a directory opened with pcv_octree_open_dir
(reference: stream_points_for_query_in_node -> Octree::points_in_node -> NodeIterator over node files,
src/iterator.rs:185-223, src/octree/mod.rs:285-307, src/read_write/node_iterator.rs:24-119), meta.pb of every version
the reference still reads (9..13, src/octree/mod.rs:156-215) written with the REAL protobuf runtime, and the
rejection of malformed meta files."""
import math
import os

import numpy as np
import pytest

import meta_proto
import oracle_lib as O
import point_cloud_viewer_amd as pcv
from point_cloud_viewer_amd import synthetic

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = pcv.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def disk_scene(ctx, tmp_path_factory):
    """A directory written by the ORACLE's literal build (not by this library), with intensity."""
    x, y, z, rgb, bmin, bmax = synthetic.gaussian_clusters(200_000, seed=21, num_clusters=5, extent=80.0,
                                                           sigma_range=(0.3, 5.0))
    inten = (np.arange(x.size) % 199).astype(np.float32)
    d = tmp_path_factory.mktemp("disk") / "octree"
    with O.max_points_per_node(1500):
        O.build_literal_dir(d, 0.001, bmin, bmax, x, y, z, rgb, inten, threads=4)
        want = O.load_dir(d)
    return dict(dir=d, bmin=bmin, bmax=bmax, oracle=want, n=x.size)


def _frusta(rng, bmin, bmax, n):
    out = []
    for _ in range(n):
        eye = rng.uniform(bmin - 10, bmax + 10)
        q = rng.normal(size=4)
        q = q / math.sqrt(float(((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]) + q[3] * q[3]))
        out.append(O.frustum_new(eye, q, O.perspective3_new(1.0, 1.2, 0.1, 100.0)))
    return out


def test_query_points_on_an_octree_opened_from_disk(ctx, disk_scene):
    """Same check as test_query_points_batched, but nothing was built in this process: decode-on-load from node files."""
    sc = disk_scene
    tree = ctx.open_dir(sc["dir"])
    on = sc["oracle"].nodes
    names = tree.node_names()
    assert set(names) == set(on) and tree.num_points == sc["n"]
    rng = np.random.default_rng(4)
    fr = _frusta(rng, sc["bmin"], sc["bmax"], 5)
    obb = (sc["bmin"] + 35, O.quat_from_axis_angle([0.0, 1.0, 0.0], 0.7), [25.0, 18.0, 12.0])
    shapes = [("frustum2", *fr[i]) for i in range(5)] + [("obb", *obb), ("aabb", sc["bmin"] + 8, sc["bmin"] + 50), ("all",)]
    kinds = [(O.SHAPE_FRUSTUM2, np.concatenate(fr[i])) for i in range(5)]
    kinds += [(O.SHAPE_OBB, list(obb[0]) + list(obb[1]) + list(obb[2])),
              (O.SHAPE_AABB, list(sc["bmin"] + 8) + list(sc["bmin"] + 50)), (O.SHAPE_ALL, None)]
    prepared = ctx.shapes(shapes)
    nonempty = 0
    for i, (kind, params) in enumerate(kinds):
        for interval in (None, (15.0, 150.0)):
            got = tree.query_points(prepared, i, interval=interval)
            wx, wy, wz, wrgb, wint = [], [], [], [], []
            for name in O.nodes_in_location(sc["bmin"], sc["bmax"], on, kind, params):
                nd = on[name]
                if nd["num_points"] == 0:
                    continue
                info = tree.node(names.index(name))
                px, py, pz = O.decode_positions(nd["encoding"], info.cube_min, info.cube_edge, nd["xyz"])
                inten = np.frombuffer(nd["intensity"], dtype=np.float32)
                keep = O.cull_points(kind, params, px, py, pz, inten if interval else None, interval).astype(bool)
                wx.append(px[keep]); wy.append(py[keep]); wz.append(pz[keep])
                wrgb.append(np.frombuffer(nd["rgb"], dtype=np.uint8).reshape(-1, 3)[keep])
                wint.append(inten[keep])
            cat = lambda parts, dt: np.concatenate(parts) if parts else np.zeros(0, dtype=dt)
            assert got["count"] == sum(len(p) for p in wx), (i, interval)
            assert np.array_equal(got["x"], cat(wx, np.float64)) and np.array_equal(got["y"], cat(wy, np.float64))
            assert np.array_equal(got["z"], cat(wz, np.float64))
            assert np.array_equal(got["rgb"].reshape(-1, 3), cat(wrgb, np.uint8).reshape(-1, 3))
            assert np.array_equal(got["intensity"], cat(wint, np.float32))
            nonempty += got["count"] > 0
    assert nonempty >= 6
    # per-node keep mask on a disk node (pcv_cull_node_points) against the oracle's contains()
    big = max(range(tree.num_nodes), key=lambda k: tree.node(k).num_points)
    nd = on[names[big]]
    info = tree.node(big)
    px, py, pz = O.decode_positions(nd["encoding"], info.cube_min, info.cube_edge, nd["xyz"])
    keep, kept = tree.cull_node_points(prepared, 6, big)
    want = O.cull_points(kinds[6][0], kinds[6][1], px, py, pz)
    assert np.array_equal(keep, want) and kept == int(want.sum())
    # stream_points_for_query_in_node: that node's points through the FilteredIterator, with an interval
    inten = np.frombuffer(nd["intensity"], dtype=np.float32)
    wi = O.cull_points(kinds[6][0], kinds[6][1], px, py, pz, inten, (15.0, 150.0)).astype(bool)
    got = tree.query_points(prepared, 6, interval=(15.0, 150.0), node=big)
    assert got["count"] == int(wi.sum()) and np.array_equal(got["x"], px[wi]) and np.array_equal(got["z"], pz[wi])
    assert np.array_equal(got["rgb"].reshape(-1, 3), np.frombuffer(nd["rgb"], dtype=np.uint8).reshape(-1, 3)[wi])
    assert np.array_equal(got["intensity"], inten[wi])
    tree.free()


def test_truncated_node_file_is_an_error_not_garbage(ctx, disk_scene, tmp_path):
    import shutil
    d = tmp_path / "copy"
    shutil.copytree(disk_scene["dir"], d)
    victim = sorted(p for p in os.listdir(d) if p.endswith(".xyz"))[-1]
    data = (d / victim).read_bytes()
    (d / victim).write_bytes(data[:-3])
    tree = ctx.open_dir(d)
    with pytest.raises(pcv.PcvError) as e:
        tree.query_points(ctx.shapes([("all",)]), 0)
    assert e.value.code == pcv.PCV_E_IO and "num_points" in str(e.value)
    tree.free()


def _write_meta_version(path, version, bmin, bmax, resolution, nodes):
    """meta.pb as the tools of `version` wrote it (src/lib.rs:41-48), serialised by the protobuf runtime."""
    cls = meta_proto.classes()
    m = cls["Meta"]()
    m.version = version

    def fill_box(box):
        if version <= 10:  # Vector3f min/max (version 10 -> 11 changed them to doubles)
            box.deprecated_min.x, box.deprecated_min.y, box.deprecated_min.z = [float(v) for v in bmin]
            box.deprecated_max.x, box.deprecated_max.y, box.deprecated_max.z = [float(v) for v in bmax]
        else:
            box.min.x, box.min.y, box.min.z = [float(v) for v in bmin]
            box.max.x, box.max.y, box.max.z = [float(v) for v in bmax]

    def fill_node(dst, level, index, npts, enc):
        dst.position_encoding = enc
        dst.num_points = npts
        if version == 9:  # level (u8) + index (u64) instead of high/low
            dst.id.deprecated_level, dst.id.deprecated_index = level, index
            dst.id.SetInParent()
        else:
            hi, lo = meta_proto.node_id(level, index)
            dst.id.high, dst.id.low = hi, lo
            dst.id.SetInParent()

    if version <= 11:
        fill_box(m.bounding_box)
        m.deprecated_resolution = resolution
        for nd in nodes:
            fill_node(m.deprecated_nodes.add(), *nd)
    else:
        fill_box(m.octree.deprecated_bounding_box if version == 12 else m.bounding_box)
        m.octree.resolution = resolution
        m.octree.SetInParent()
        for nd in nodes:
            fill_node(m.octree.nodes.add(), *nd)
    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, "meta.pb"), "wb") as f:
        f.write(m.SerializeToString())


@pytest.mark.parametrize("version", [9, 10, 11, 12, 13])
def test_open_dir_reads_every_meta_version_the_reference_reads(ctx, tmp_path, version):
    # bounds exactly representable in f32, so that the Vector3f versions carry the same box
    bmin, bmax = np.array([-64.0, 8.0, 0.5]), np.array([64.0, 72.0, 96.5])
    nodes = [(0, 0, 10, 3), (1, 3, 0, 3), (1, 5, 77, 2), (2, 0o52, 5, 2), (3, 0o527, 1, 1)]
    _write_meta_version(tmp_path / "o", version, bmin, bmax, 0.25, nodes)
    t = ctx.open_dir(tmp_path / "o")
    m = t.meta()
    assert np.array_equal(m["bbox_min"], bmin) and np.array_equal(m["bbox_max"], bmax) and m["resolution"] == 0.25
    assert t.node_names() == ["r", "r3", "r5", "r52", "r527"]
    assert [t.node(i).num_points for i in range(5)] == [10, 0, 77, 5, 1]
    assert [t.node(i).encoding for i in range(5)] == [3, 3, 2, 2, 1]
    # NodeId::find_bounding_cube (node.rs:157-172) through the oracle
    for i, (level, index, _, _) in enumerate(nodes):
        hi, lo = meta_proto.node_id(level, index)
        mn, edge = O.find_bounding_cube(hi, lo, bmin, 128.0)
        assert tuple(t.node(i).cube_min) == tuple(mn) and t.node(i).cube_edge == edge
    t.free()


def test_open_dir_rejects_what_the_reference_rejects(ctx, tmp_path):
    bmin, bmax = [0.0, 0, 0], [1.0, 1, 1]
    for version in (8, 14):
        _write_meta_version(tmp_path / f"v{version}", version, bmin, bmax, 0.1, [(0, 0, 1, 1)])
        with pytest.raises(pcv.PcvError) as e:
            ctx.open_dir(tmp_path / f"v{version}")
        assert "InvalidVersion" in str(e.value)
    # PositionEncoding INVALID (codec.rs:50-53)
    _write_meta_version(tmp_path / "enc0", 13, bmin, bmax, 0.1, [(0, 0, 1, 0)])
    with pytest.raises(pcv.PcvError):
        ctx.open_dir(tmp_path / "enc0")
    # version 13 without the octree sub-message (octree/mod.rs:179-181)
    cls = meta_proto.classes()
    m = cls["Meta"]()
    m.version = 13
    os.makedirs(tmp_path / "nooct")
    (tmp_path / "nooct" / "meta.pb").write_bytes(m.SerializeToString())
    with pytest.raises(pcv.PcvError) as e:
        ctx.open_dir(tmp_path / "nooct")
    assert "No octree meta" in str(e.value)
    # a level no NodeId can name (level byte 200): rejected instead of shifting a u128 out of range
    m = cls["Meta"]()
    m.version = 13
    nd = m.octree.nodes.add()
    nd.position_encoding, nd.num_points, nd.id.high, nd.id.low = 1, 1, 200 << 56, 5
    os.makedirs(tmp_path / "deep")
    (tmp_path / "deep" / "meta.pb").write_bytes(m.SerializeToString())
    with pytest.raises(pcv.PcvError) as e:
        ctx.open_dir(tmp_path / "deep")
    assert e.value.code == pcv.PCV_E_INVALID
    # num_points is untrusted: a negative count or counts that add up past any addressable blob would turn into raw
    # write offsets when the node files are loaded (safe Rust cannot do that; here it must be refused at open time)
    for tag, counts in (("neg", [-2, 7]), ("huge", [1 << 62, 5]), ("sum", [(1 << 56) // 40] * 3)):
        _write_meta_version(tmp_path / tag, 13, bmin, bmax, 0.1, [(0, 0, counts[0], 1)] + [(1, k, c, 1) for k, c in enumerate(counts[1:])])
        with pytest.raises(pcv.PcvError) as e:
            ctx.open_dir(tmp_path / tag)
        assert e.value.code == pcv.PCV_E_INVALID and "num_points" in str(e.value), tag
    # garbage
    os.makedirs(tmp_path / "junk")
    (tmp_path / "junk" / "meta.pb").write_bytes(b"\xff" * 37)
    with pytest.raises(pcv.PcvError):
        ctx.open_dir(tmp_path / "junk")


def test_stream_hand_off_with_torch(ctx):
    """pcv_ctx_wait_stream / pcv_ctx_signal_stream: inputs produced on torch's stream right before the call, outputs
    consumed on torch's stream right after an asynchronous copy — without a host synchronisation in between."""
    import torch
    dev = torch.device("cuda", 0)
    side = torch.cuda.Stream(device=dev)
    n = 4_000_000
    for rep in range(5):
        with torch.cuda.stream(side):
            base = torch.full((n,), float(rep), dtype=torch.float64, device=dev)
            for _ in range(20):  # keep the side stream busy so that an unordered read would see stale data
                base = base * 1.0000001 + 0.5
            x = base.clone()
            ctx.wait_stream(side.cuda_stream)
        bmin, bmax = ctx.aabb_reduce(x, x, x)
        assert bmin[0] == float(x.min().item()) and bmax[0] == float(x.max().item())
    # the default stream (handle 0)
    y = torch.arange(n, dtype=torch.float64, device=dev) * 2.0 - 7.0
    ctx.wait_torch()
    bmin, bmax = ctx.aabb_reduce(y, y, y)
    assert bmin[0] == -7.0 and bmax[0] == 2.0 * (n - 1) - 7.0
    ctx.signal_torch()
