//! 1:1 veneer over include/pcv_hip.h that keeps the reference crate's public surface for the hot path:
//! `build_octree` (src/octree/generation.rs:289-295), `Octree::get_visible_nodes` (src/octree/mod.rs:228), and the
//! `PointCloud` trait (src/iterator.rs:169-206: `nodes_in_location`, `encoding_for_node`, `points_in_node`,
//! `bounding_box`, `stream_points_for_query_in_node`) as `HipOctree`, which `ParallelIterator` / `PointCloudClient`
//! take unchanged. All logic lives behind the C ABI; this file only marshals `PointsBatch` / nalgebra types into plain
//! pointers. Uncompiled here (no Rust toolchain in the build container) — see INTEGRATION.md for how it slots into the
//! reference workspace and for the four accessor one-liners it needs on `Frustum` / `Obb` (their fields are private).
use nalgebra::{Matrix4, Point3, Vector3};
use point_viewer::data_provider::OnDiskDataProvider;
use point_viewer::errors::{ErrorKind, Result};
use point_viewer::geometry::Aabb;
use point_viewer::iterator::{PointCloud, PointLocation, PointQuery};
use point_viewer::octree::{NodeId, Octree};
use point_viewer::read_write::{Encoding, NodeIterator, PositionEncoding};
use point_viewer::{AttributeData, NumberOfPoints, PointsBatch};
use std::collections::{BTreeMap, HashMap};
use std::ffi::{CStr, CString};
use std::os::raw::{c_char, c_double, c_float, c_int, c_void};
use std::path::Path;
use std::sync::Mutex;

#[repr(C)]
pub struct PcvPoints {
    n: u64,
    x: *const c_double,
    y: *const c_double,
    z: *const c_double,
    color: *const u8,
    color_stride: u32,
    intensity: *const c_float,
    mem: i32,
}

#[repr(C)]
pub struct PcvBuildParams {
    resolution: c_double,
    bbox_min: [c_double; 3],
    bbox_max: [c_double; 3],
    max_points_per_node: u32,
    flags: u32,
}

#[repr(C)]
pub struct PcvShape {
    kind: i32,
    reserved: i32,
    params: [c_double; 32],
}

#[repr(C)]
#[derive(Default, Clone, Copy)]
pub struct PcvNodeInfo {
    id_high: u64,
    id_low: u64,
    num_points: i64,
    level: u32,
    encoding: u32,
    cube_min: [c_double; 3],
    cube_edge: c_double,
    xyz_offset: u64,
    point_offset: u64,
}

#[allow(non_camel_case_types)]
type pcv_ctx = c_void;
#[allow(non_camel_case_types)]
type pcv_octree = c_void;
#[allow(non_camel_case_types)]
type pcv_shapes = c_void;
#[allow(non_camel_case_types)]
type pcv_ingest = c_void;

extern "C" {
    fn pcv_abi_version() -> c_int;
    fn pcv_ctx_create(device: c_int, stream: *mut c_void, out: *mut *mut pcv_ctx) -> c_int;
    fn pcv_ctx_destroy(ctx: *mut pcv_ctx);
    fn pcv_last_error(ctx: *const pcv_ctx) -> *const c_char;
    #[allow(dead_code)]
    fn pcv_build_octree(ctx: *mut pcv_ctx, params: *const PcvBuildParams, points: *const PcvPoints, out: *mut *mut pcv_octree) -> c_int;
    // streaming batch ingest: `impl Iterator<Item = PointsBatch>` of generation.rs:289-295, one batch per call
    fn pcv_ingest_begin(ctx: *mut pcv_ctx, num_points_hint: u64, has_intensity: c_int, out: *mut *mut pcv_ingest) -> c_int;
    fn pcv_ingest_append(ingest: *mut pcv_ingest, xyz: *const c_double, rgb: *const u8, intensity: *const c_float, n: u64) -> c_int;
    fn pcv_ingest_finish(ingest: *mut pcv_ingest, params: *const PcvBuildParams, out: *mut *mut pcv_octree) -> c_int;
    fn pcv_ingest_abort(ingest: *mut pcv_ingest);
    fn pcv_build_octree_from_ply(ctx: *mut pcv_ctx, params: *const PcvBuildParams, path: *const c_char, with_intensity: c_int, out: *mut *mut pcv_octree) -> c_int;
    fn pcv_octree_write_dir(t: *mut pcv_octree, directory: *const c_char) -> c_int;
    fn pcv_octree_open_dir(ctx: *mut pcv_ctx, directory: *const c_char, out: *mut *mut pcv_octree) -> c_int;
    fn pcv_octree_num_nodes(t: *const pcv_octree) -> u64;
    fn pcv_octree_node(t: *const pcv_octree, i: u64, out: *mut PcvNodeInfo) -> c_int;
    fn pcv_octree_free(t: *mut pcv_octree);
    fn pcv_shapes_create(ctx: *mut pcv_ctx, shapes: *const PcvShape, count: u32, out: *mut *mut pcv_shapes) -> c_int;
    fn pcv_shapes_free(s: *mut pcv_shapes);
    fn pcv_visible_nodes(ctx: *mut pcv_ctx, frusta: *const pcv_shapes, t: *mut pcv_octree, capacity: u32, counts: *mut u32, node_indices: *mut u32, status: *mut i32) -> c_int;
    // per frustum the nodes whose Relation is not Out, with relation and relative_size_on_screen (octree/mod.rs:119-139, 261-272)
    #[allow(dead_code)]
    fn pcv_cull_nodes_sparse(ctx: *mut pcv_ctx, shapes: *const pcv_shapes, t: *mut pcv_octree, capacity: u32, counts: *mut u32, node_indices: *mut u32, relation: *mut u8, size_on_screen: *mut c_double) -> c_int;
    fn pcv_nodes_in_location(ctx: *mut pcv_ctx, shapes: *const pcv_shapes, t: *mut pcv_octree, capacity: u32, counts: *mut u32, node_indices: *mut u32) -> c_int;
    fn pcv_query_node_points(ctx: *mut pcv_ctx, shapes: *const pcv_shapes, shape_index: u32, t: *mut pcv_octree, node: u64, interval: *const c_double, capacity: u64, mem: c_int, x: *mut c_double, y: *mut c_double, z: *mut c_double, rgb: *mut u8, intensity: *mut c_float, count: *mut u64) -> c_int;
    fn pcv_octree_has_intensity(t: *const pcv_octree) -> c_int;
}

pub struct HipContext(*mut pcv_ctx);

impl HipContext {
    pub fn new(device: i32) -> Result<Self, String> {
        // include/pcv_hip.h PCV_ABI_VERSION these declarations were written against (2: pcv_ingest_*, PCV_STAGE_SORT_SECOND)
        if unsafe { pcv_abi_version() } != 2 {
            return Err(format!("libpcv_hip.so has ABI version {}, this veneer expects 2", unsafe { pcv_abi_version() }));
        }
        let mut ctx = std::ptr::null_mut();
        match unsafe { pcv_ctx_create(device, std::ptr::null_mut(), &mut ctx) } {
            0 => Ok(HipContext(ctx)),
            rc => Err(format!("pcv_ctx_create failed: {}", rc)),
        }
    }
    fn check(&self, rc: c_int) {
        if rc != 0 {
            // The reference's builder panics on every error (generation.rs:99,177,376); keep that contract.
            let msg = unsafe { CStr::from_ptr(pcv_last_error(self.0)) }.to_string_lossy().into_owned();
            panic!("pcv_hip error {}: {}", rc, msg);
        }
    }
}

impl Drop for HipContext {
    fn drop(&mut self) {
        unsafe { pcv_ctx_destroy(self.0) }
    }
}

thread_local! {
    /// One context per host thread for the lifetime of the thread (a `pcv_ctx` is bound to one device and one stream
    /// and is not thread-safe; creating one costs a stream, events and the first pinned allocations).
    static CONTEXT: HipContext = HipContext::new(0).expect("no MI355X visible (there is no CPU fallback)");
}

/// Same signature and behaviour as `point_viewer::octree::build_octree` (generation.rs:289-295). The batches go to the
/// device ONE AT A TIME and AS THEY ARE (pcv_ingest_append): `batch.position` is a `Vec<Point3<f64>>` — `Point3<f64>` is
/// `#[repr(C)]` over `[f64; 3]`, so the vector is n x 3 contiguous doubles — "color" a `Vec<Vector3<u8>>` (n x 3 bytes),
/// "intensity" a `Vec<f32>`. The library copies the three slices into its pinned ring, queues one DMA and one kernel that
/// transposes AoS -> SoA on the device, and returns; the iterator produces its next batch meanwhile. No whole-cloud host
/// vector exists at any time (host memory: the ring, 3 x 32 MiB), nothing is transposed on the host.
pub fn build_octree(
    output_directory: impl AsRef<Path>,
    resolution: f64,
    bounding_box: Aabb,
    input: impl Iterator<Item = PointsBatch> + NumberOfPoints + Send,
    attributes: &[&str],
) {
    let want_intensity = attributes.contains(&"intensity");
    CONTEXT.with(|ctx| {
        let mut ingest = std::ptr::null_mut();
        ctx.check(unsafe { pcv_ingest_begin(ctx.0, input.num_points() as u64, want_intensity as c_int, &mut ingest) });
        for batch in input {
            let color = match batch.attributes.get("color") {
                Some(AttributeData::U8Vec3(c)) => c.as_ptr() as *const u8,
                _ => panic!("color attribute (U8Vec3) is required"), // on_disk.rs:20-22
            };
            let intensity = if want_intensity {
                match batch.attributes.get("intensity") {
                    Some(AttributeData::F32(i)) => i.as_ptr(),
                    _ => panic!("intensity requested but missing"), // generation.rs:167-177 unwrap()
                }
            } else {
                std::ptr::null()
            };
            let rc = unsafe { pcv_ingest_append(ingest, batch.position.as_ptr() as *const c_double, color, intensity, batch.position.len() as u64) };
            if rc != 0 {
                unsafe { pcv_ingest_abort(ingest) };
                ctx.check(rc);
            }
            // `batch` is dropped here: the library has copied it into pinned memory before returning
        }
        let params = PcvBuildParams {
            resolution,
            bbox_min: [bounding_box.min().x, bounding_box.min().y, bounding_box.min().z],
            bbox_max: [bounding_box.max().x, bounding_box.max().y, bounding_box.max().z],
            max_points_per_node: 0,
            flags: 0,
        };
        let mut tree = std::ptr::null_mut();
        ctx.check(unsafe { pcv_ingest_finish(ingest, &params, &mut tree) }); // consumes the ingest whatever it returns
        let dir = CString::new(output_directory.as_ref().to_str().unwrap()).unwrap();
        ctx.check(unsafe { pcv_octree_write_dir(tree, dir.as_ptr()) });
        unsafe { pcv_octree_free(tree) };
    });
}

/// Same signature and behaviour as `point_viewer::octree::build_octree_from_file` (generation.rs:272-287), which is what
/// `src/bin/build_octree.rs:47-52` calls: the PLY's vertex records go to the device as they are in the file and are
/// decoded there (cast to f64 + `comment offset`, ply.rs:488-493), the bounding box is computed on the device
/// (find_bounding_box, generation.rs:256-270), then build + directory write. Panics like the reference on any error,
/// including a PLY without the `intensity` the attribute list asks for.
pub fn build_octree_from_file(output_directory: impl AsRef<Path>, resolution: f64, filename: impl AsRef<Path>, attributes: &[&str]) {
    CONTEXT.with(|ctx| {
        let params = PcvBuildParams { resolution, bbox_min: [0.0; 3], bbox_max: [0.0; 3], max_points_per_node: 0, flags: 0 };
        let file = CString::new(filename.as_ref().to_str().unwrap()).unwrap();
        let mut tree = std::ptr::null_mut();
        let with_intensity = attributes.contains(&"intensity") as c_int;
        ctx.check(unsafe { pcv_build_octree_from_ply(ctx.0, &params, file.as_ptr(), with_intensity, &mut tree) });
        let dir = CString::new(output_directory.as_ref().to_str().unwrap()).unwrap();
        ctx.check(unsafe { pcv_octree_write_dir(tree, dir.as_ptr()) });
        unsafe { pcv_octree_free(tree) };
    });
}

/// `Octree::get_visible_nodes` (octree/mod.rs:228-283) for one matrix over an octree directory.
pub fn get_visible_nodes(ctx: &HipContext, directory: &Path, projection_matrix: &Matrix4<f64>) -> Vec<NodeId> {
    let dir = CString::new(directory.to_str().unwrap()).unwrap();
    let mut tree = std::ptr::null_mut();
    ctx.check(unsafe { pcv_octree_open_dir(ctx.0, dir.as_ptr(), &mut tree) });
    let mut shape = PcvShape { kind: 2, reserved: 0, params: [0.0; 32] };
    shape.params[..16].copy_from_slice(projection_matrix.as_slice()); // nalgebra storage is column-major
    let mut shapes = std::ptr::null_mut();
    ctx.check(unsafe { pcv_shapes_create(ctx.0, &shape, 1, &mut shapes) });
    let m = unsafe { pcv_octree_num_nodes(tree) } as usize;
    let (mut count, mut status) = (0u32, 0i32);
    let mut idx = vec![0u32; m.max(1)];
    ctx.check(unsafe { pcv_visible_nodes(ctx.0, shapes, tree, m as u32, &mut count, idx.as_mut_ptr(), &mut status) });
    assert!(status == 0, "Invalid projection matrix."); // octree/mod.rs:230 .expect(...)
    let ids = idx[..count as usize]
        .iter()
        .map(|&i| {
            let mut info = PcvNodeInfo::default();
            unsafe { pcv_octree_node(tree, i as u64, &mut info) };
            NodeId::from_level_index(info.level as u8, ((info.id_high as u128 & 0x00ff_ffff_ffff_ffff) << 64) | info.id_low as u128)
        })
        .collect();
    unsafe {
        pcv_shapes_free(shapes);
        pcv_octree_free(tree);
    }
    ids
}


// ------------------------------------------------------------------------------------------------------------------
// PointCloud over an octree directory: what ParallelIterator (src/iterator.rs:226-333), PointCloudClient
// (point_cloud_client/src/lib.rs:27-50) and xray tile generation (xray/src/generation.rs:464-513) call.
// ------------------------------------------------------------------------------------------------------------------

/// An octree on disk served by the GPU library. The reference `Octree` is kept alongside for the two things that stay
/// on the host: `points_in_node` (a plain `NodeIterator` over one node's files) and locations the library has no
/// kernel for (S2 cells, web-mercator rectangles: third-party math).
pub struct HipOctree {
    ctx: HipContext,
    tree: *mut pcv_octree,
    lock: Mutex<()>, // a pcv_ctx is not thread-safe; ParallelIterator calls from several workers
    ids: Vec<NodeId>,
    index_of: HashMap<NodeId, u64>,
    infos: Vec<PcvNodeInfo>,
    inner: Octree,
}

// The raw handles are only touched under `lock`.
unsafe impl Send for HipOctree {}
unsafe impl Sync for HipOctree {}

impl HipOctree {
    pub fn from_directory(directory: impl AsRef<Path>) -> Result<Self> {
        let ctx = HipContext::new(0).map_err(|e| ErrorKind::InvalidInput(e))?;
        let dir = CString::new(directory.as_ref().to_str().unwrap()).unwrap();
        let mut tree = std::ptr::null_mut();
        if unsafe { pcv_octree_open_dir(ctx.0, dir.as_ptr(), &mut tree) } != 0 {
            let msg = unsafe { CStr::from_ptr(pcv_last_error(ctx.0)) }.to_string_lossy().into_owned();
            return Err(ErrorKind::InvalidInput(msg).into());
        }
        let m = unsafe { pcv_octree_num_nodes(tree) };
        let mut ids = Vec::with_capacity(m as usize);
        let mut infos = Vec::with_capacity(m as usize);
        let mut index_of = HashMap::with_capacity(m as usize);
        for i in 0..m {
            let mut info = PcvNodeInfo::default();
            unsafe { pcv_octree_node(tree, i, &mut info) };
            let id = NodeId::from_level_index(info.level as u8, ((info.id_high as u128 & 0x00ff_ffff_ffff_ffff) << 64) | info.id_low as u128);
            index_of.insert(id, i);
            ids.push(id);
            infos.push(info);
        }
        let inner = Octree::from_data_provider(Box::new(OnDiskDataProvider { directory: directory.as_ref().to_path_buf() }))?;
        Ok(HipOctree { ctx, tree, lock: Mutex::new(()), ids, index_of, infos, inner })
    }

    /// PointLocation -> pcv_shape (include/pcv_hip.h). None: a location the library has no kernel for.
    /// Needs `Frustum::clip_from_query()/query_from_clip()` and `Obb::query_from_obb()/half_extent()` accessors on
    /// the reference types (their fields are private; INTEGRATION.md lists the four one-liners).
    fn shape_of(location: &PointLocation) -> Option<PcvShape> {
        let mut s = PcvShape { kind: 0, reserved: 0, params: [0.0; 32] };
        match location {
            PointLocation::AllPoints => s.kind = 0,
            PointLocation::Aabb(b) => {
                s.kind = 1;
                s.params[..3].copy_from_slice(&[b.min().x, b.min().y, b.min().z]);
                s.params[3..6].copy_from_slice(&[b.max().x, b.max().y, b.max().z]);
            }
            PointLocation::Frustum(f) => {
                s.kind = 4; // PCV_SHAPE_FRUSTUM_WITH_INVERSE: exactly the two matrices Frustum::new stored
                s.params[..16].copy_from_slice(f.clip_from_query().as_slice());
                s.params[16..32].copy_from_slice(f.query_from_clip().as_slice());
            }
            PointLocation::Obb(o) => {
                s.kind = 3;
                let iso = o.query_from_obb();
                let (t, q, h) = (iso.translation.vector, iso.rotation.coords, o.half_extent());
                s.params[..3].copy_from_slice(&[t.x, t.y, t.z]);
                s.params[3..7].copy_from_slice(&[q.x, q.y, q.z, q.w]); // i j k w
                s.params[7..10].copy_from_slice(&[h.x, h.y, h.z]);
            }
            PointLocation::S2Cells(_) | PointLocation::WebMercatorRect(_) => return None,
        }
        Some(s)
    }

    fn with_shape<R>(&self, shape: &PcvShape, f: impl FnOnce(*mut pcv_shapes) -> R) -> R {
        let mut shapes = std::ptr::null_mut();
        self.ctx.check(unsafe { pcv_shapes_create(self.ctx.0, shape, 1, &mut shapes) });
        let r = f(shapes);
        unsafe { pcv_shapes_free(shapes) };
        r
    }
}

impl Drop for HipOctree {
    fn drop(&mut self) {
        unsafe { pcv_octree_free(self.tree) } // before the context (field order: ctx is dropped after this body)
    }
}

impl PointCloud for HipOctree {
    type Id = NodeId;

    /// src/octree/mod.rs:329-331 + octree_iterator.rs: breadth first, a node is reported iff its cube is not Out.
    fn nodes_in_location(&self, location: &PointLocation) -> Vec<NodeId> {
        let shape = match Self::shape_of(location) {
            Some(s) => s,
            None => return self.inner.nodes_in_location(location),
        };
        let _g = self.lock.lock().unwrap();
        self.with_shape(&shape, |shapes| {
            let cap = self.ids.len().max(1);
            let mut count = 0u32;
            let mut idx = vec![0u32; cap];
            self.ctx.check(unsafe { pcv_nodes_in_location(self.ctx.0, shapes, self.tree, cap as u32, &mut count, idx.as_mut_ptr()) });
            idx[..count as usize].iter().map(|&i| self.ids[i as usize]).collect()
        })
    }

    /// src/octree/mod.rs:76-84
    fn encoding_for_node(&self, id: NodeId) -> Encoding {
        let info = &self.infos[self.index_of[&id] as usize];
        let enc = match info.encoding {
            1 => PositionEncoding::Uint8,
            2 => PositionEncoding::Uint16,
            3 => PositionEncoding::Float32,
            _ => PositionEncoding::Float64,
        };
        Encoding::ScaledToCube(Point3::new(info.cube_min[0], info.cube_min[1], info.cube_min[2]), info.cube_edge, enc)
    }

    fn points_in_node(&self, attributes: &[&str], node_id: NodeId, batch_size: usize) -> Result<NodeIterator> {
        self.inner.points_in_node(attributes, node_id, batch_size)
    }

    fn bounding_box(&self) -> &Aabb {
        self.inner.bounding_box()
    }

    /// src/iterator.rs:185-205: decode + FilteredIterator + retain of ONE node on the GPU (decode on load from the
    /// node's bytes, keep mask, stable compaction), handed to the callback in batches of `batch_size`; an `Err` from
    /// the callback aborts the stream like in the reference.
    fn stream_points_for_query_in_node<F>(&self, query: &PointQuery, node_id: NodeId, batch_size: usize, mut callback: F) -> Result<()>
    where
        F: FnMut(PointsBatch) -> Result<()>,
    {
        let only_intensity = query.filter_intervals.keys().all(|k| *k == "intensity");
        let shape = match (Self::shape_of(&query.location), only_intensity) {
            (Some(s), true) => s,
            _ => {
                // no kernel for this location / an interval on another attribute: the reference's host path
                let it = self.inner.points_in_node(&query.attributes, node_id, batch_size)?;
                return point_viewer::iterator::stream_dispatch(&query.location, &query.filter_intervals, it, callback);
            }
        };
        let node = self.index_of[&node_id];
        let cap = self.infos[node as usize].num_points as usize;
        if cap == 0 {
            return Ok(());
        }
        let interval = query.filter_intervals.get("intensity").map(|iv| [iv.lower_bound, iv.upper_bound]);
        let (mut x, mut y, mut z) = (vec![0f64; cap], vec![0f64; cap], vec![0f64; cap]);
        let mut rgb = vec![0u8; 3 * cap];
        let want_intensity = query.attributes.contains(&"intensity");
        let mut inten = vec![0f32; if want_intensity { cap } else { 0 }];
        let mut count = 0u64;
        {
            let _g = self.lock.lock().unwrap();
            let has_int = unsafe { pcv_octree_has_intensity(self.tree) } != 0;
            self.with_shape(&shape, |shapes| {
                self.ctx.check(unsafe {
                    pcv_query_node_points(
                        self.ctx.0, shapes, 0, self.tree, node,
                        interval.as_ref().map_or(std::ptr::null(), |iv| iv.as_ptr()),
                        cap as u64, 0, x.as_mut_ptr(), y.as_mut_ptr(), z.as_mut_ptr(), rgb.as_mut_ptr(),
                        if want_intensity && has_int { inten.as_mut_ptr() } else { std::ptr::null_mut() }, &mut count,
                    )
                })
            });
        }
        let n = count as usize;
        let mut at = 0;
        while at < n {
            let end = (at + batch_size).min(n);
            let position = (at..end).map(|i| Point3::new(x[i], y[i], z[i])).collect();
            let mut attributes = BTreeMap::new();
            if query.attributes.contains(&"color") {
                attributes.insert("color".to_string(), AttributeData::U8Vec3((at..end).map(|i| Vector3::new(rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2])).collect()));
            }
            if want_intensity {
                attributes.insert("intensity".to_string(), AttributeData::F32(inten[at..end].to_vec()));
            }
            callback(PointsBatch { position, attributes })?;
            at = end;
        }
        Ok(())
    }
}

impl HipOctree {
    /// `Octree::get_visible_nodes` (octree/mod.rs:228-283) on the open tree: node ids in the order the reference's
    /// BinaryHeap pops them. Panics like the reference on a matrix that cannot be inverted.
    pub fn get_visible_nodes(&self, projection_matrix: &Matrix4<f64>) -> Vec<NodeId> {
        let mut shape = PcvShape { kind: 2, reserved: 0, params: [0.0; 32] };
        shape.params[..16].copy_from_slice(projection_matrix.as_slice()); // nalgebra storage is column-major
        let _g = self.lock.lock().unwrap();
        self.with_shape(&shape, |shapes| {
            let m = self.ids.len().max(1);
            let (mut count, mut status) = (0u32, 0i32);
            let mut idx = vec![0u32; m];
            self.ctx.check(unsafe { pcv_visible_nodes(self.ctx.0, shapes, self.tree, m as u32, &mut count, idx.as_mut_ptr(), &mut status) });
            assert!(status == 0, "Invalid projection matrix."); // octree/mod.rs:230 .expect(...)
            idx[..count as usize].iter().map(|&i| self.ids[i as usize]).collect()
        })
    }
}
