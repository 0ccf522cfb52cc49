//! 1:1 veneer over include/pcv_hip.h that keeps the reference crate's public surface for the hot path:
//! `build_octree` (src/octree/generation.rs:289-295), `Octree::get_visible_nodes` (src/octree/mod.rs:228),
//! `PointCulling::contains` batched (src/math/base.rs:5-7). All logic lives behind the C ABI; this file only
//! marshals `PointsBatch` / nalgebra types into plain pointers. Uncompiled here (no Rust toolchain in the
//! build container) — see INTEGRATION.md for how it slots into the reference workspace.
use nalgebra::Matrix4;
use point_viewer::geometry::Aabb;
use point_viewer::octree::NodeId;
use point_viewer::{AttributeData, NumberOfPoints, PointsBatch};
use std::ffi::{CStr, CString};
use std::os::raw::{c_char, c_double, c_float, c_int, c_void};
use std::path::Path;

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

extern "C" {
    fn pcv_ctx_create(device: c_int, stream: *mut c_void, out: *mut *mut pcv_ctx) -> c_int;
    fn pcv_ctx_destroy(ctx: *mut pcv_ctx);
    fn pcv_last_error(ctx: *const pcv_ctx) -> *const c_char;
    fn pcv_build_octree(ctx: *mut pcv_ctx, params: *const PcvBuildParams, points: *const PcvPoints, out: *mut *mut pcv_octree) -> c_int;
    fn pcv_octree_write_dir(t: *mut pcv_octree, directory: *const c_char) -> c_int;
    fn pcv_octree_open_dir(ctx: *mut pcv_ctx, directory: *const c_char, out: *mut *mut pcv_octree) -> c_int;
    fn pcv_octree_num_nodes(t: *const pcv_octree) -> u64;
    fn pcv_octree_node(t: *const pcv_octree, i: u64, out: *mut PcvNodeInfo) -> c_int;
    fn pcv_octree_free(t: *mut pcv_octree);
    fn pcv_shapes_create(ctx: *mut pcv_ctx, shapes: *const PcvShape, count: u32, out: *mut *mut pcv_shapes) -> c_int;
    fn pcv_shapes_free(s: *mut pcv_shapes);
    fn pcv_visible_nodes(ctx: *mut pcv_ctx, frusta: *const pcv_shapes, t: *mut pcv_octree, capacity: u32, counts: *mut u32, node_indices: *mut u32, status: *mut i32) -> c_int;
}

pub struct HipContext(*mut pcv_ctx);

impl HipContext {
    pub fn new(device: i32) -> Result<Self, String> {
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

/// Same signature and behaviour as `point_viewer::octree::build_octree` (generation.rs:289-295): drains the batch
/// iterator into SoA staging (one pass, no per-level files), builds on the GPU, writes the directory.
pub fn build_octree(
    output_directory: impl AsRef<Path>,
    resolution: f64,
    bounding_box: Aabb,
    input: impl Iterator<Item = PointsBatch> + NumberOfPoints + Send,
    attributes: &[&str],
) {
    let n = input.num_points();
    let (mut x, mut y, mut z) = (Vec::with_capacity(n), Vec::with_capacity(n), Vec::with_capacity(n));
    let mut rgb: Vec<u8> = Vec::with_capacity(3 * n);
    let mut intensity: Vec<f32> = Vec::new();
    let want_intensity = attributes.contains(&"intensity");
    for batch in input {
        for p in &batch.position {
            x.push(p.x);
            y.push(p.y);
            z.push(p.z);
        }
        match batch.attributes.get("color") {
            Some(AttributeData::U8Vec3(c)) => c.iter().for_each(|v| rgb.extend_from_slice(&[v.x, v.y, v.z])),
            _ => panic!("color attribute (U8Vec3) is required"),
        }
        if want_intensity {
            match batch.attributes.get("intensity") {
                Some(AttributeData::F32(i)) => intensity.extend_from_slice(i),
                _ => panic!("intensity requested but missing"), // generation.rs:167-177 unwrap()
            }
        }
    }
    let ctx = HipContext::new(0).expect("no MI355X visible");
    let params = PcvBuildParams {
        resolution,
        bbox_min: [bounding_box.min().x, bounding_box.min().y, bounding_box.min().z],
        bbox_max: [bounding_box.max().x, bounding_box.max().y, bounding_box.max().z],
        max_points_per_node: 0,
        flags: 0,
    };
    let points = PcvPoints {
        n: x.len() as u64,
        x: x.as_ptr(),
        y: y.as_ptr(),
        z: z.as_ptr(),
        color: rgb.as_ptr(),
        color_stride: 3,
        intensity: if want_intensity { intensity.as_ptr() } else { std::ptr::null() },
        mem: 0,
    };
    let mut tree = std::ptr::null_mut();
    ctx.check(unsafe { pcv_build_octree(ctx.0, &params, &points, &mut tree) });
    let dir = CString::new(output_directory.as_ref().to_str().unwrap()).unwrap();
    ctx.check(unsafe { pcv_octree_write_dir(tree, dir.as_ptr()) });
    unsafe { pcv_octree_free(tree) };
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
