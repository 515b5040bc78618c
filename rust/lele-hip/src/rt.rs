// Hand-written part of the crate (tools/rust_shim/gen.py generates ffi.rs, kernels.rs and lib.rs beside it and leaves this file alone).
//! Runtime glue between lele's host-slice API and the device library: the per-thread context, the `out: &mut Vec<T>` -> LeleBuf
//! slot registry, error handling (a non-zero status becomes the panic! lele's kernels raise), and the host-side shape arithmetic
//! of the index operators (computed exactly as the reference does, then handed to ONE strided-copy kernel).
use crate::ffi;
use crate::tensor::TensorView;
use std::cell::RefCell;
use std::collections::HashMap;
use std::ffi::CStr;
use std::os::raw::{c_int, c_void};
use std::rc::Rc;

/// element types of TensorView (src/tensor.rs:14-20); DTYPE is the LeleDType tag.  Every generic `lele::kernels` function of this
/// crate carries this bound on its element parameters IN ADDITION to upstream's (generated model code only instantiates them
/// with these five types, so a stricter bound changes nothing for it).
pub trait ElementOps: Copy + Clone + std::fmt::Debug + Default + 'static {
    const DTYPE: i32;
    fn as_f32(self) -> f32;
}
impl ElementOps for f32 {
    const DTYPE: i32 = ffi::LELE_F32;
    fn as_f32(self) -> f32 {
        self
    }
}
impl ElementOps for i64 {
    const DTYPE: i32 = ffi::LELE_I64;
    fn as_f32(self) -> f32 {
        self as f32
    }
}
impl ElementOps for i32 {
    const DTYPE: i32 = ffi::LELE_I32;
    fn as_f32(self) -> f32 {
        self as f32
    }
}
impl ElementOps for u8 {
    const DTYPE: i32 = ffi::LELE_U8;
    fn as_f32(self) -> f32 {
        self as f32
    }
}
impl ElementOps for i8 {
    const DTYPE: i32 = ffi::LELE_I8;
    fn as_f32(self) -> f32 {
        self as f32
    }
}
pub trait AsI64 {
    fn as_i64(self) -> i64;
}
impl AsI64 for f32 {
    fn as_i64(self) -> i64 {
        self as i64
    }
}
impl AsI64 for i64 {
    fn as_i64(self) -> i64 {
        self
    }
}
impl AsI64 for i32 {
    fn as_i64(self) -> i64 {
        self as i64
    }
}
impl AsI64 for u8 {
    fn as_i64(self) -> i64 {
        self as i64
    }
}
impl AsI64 for i8 {
    fn as_i64(self) -> i64 {
        self as i64
    }
}

/// A workspace slot = one LeleBuf.  lele's generated code passes `&mut ws.buf_k` (a Vec that lives in the model's workspace
/// struct for the life of the model, src/compiler/mod.rs:148-290); the Vec OBJECT's address identifies the slot.  The Vec itself
/// stays empty: it is the borrow token that ties the returned TensorView's lifetime to the slot, exactly as upstream ties it to
/// the Vec's storage.  (If a model struct is moved between calls its Vecs get new addresses and simply map to new slots.)
#[derive(Clone, Copy)]
pub struct Slot(*mut ffi::LeleBuf);
impl Slot {
    pub fn raw(&self) -> *mut ffi::LeleBuf {
        self.0
    }
    pub fn data(&self) -> *const c_void {
        unsafe { ffi::lele_hip_buf_data(self.0) as *const c_void }
    }
}

struct Runtime {
    ctx: *mut ffi::LeleCtx,
    slots: HashMap<usize, Slot>,
    scratch: Vec<Slot>,
    free: Vec<Slot>, // pooled buffers no owned view refers to any more
    prepared: HashMap<(usize, usize), PreparedWeights>,
}

/// A pooled buffer behind an OWNED result (`TensorView<'static>`: split_owned, lele::features::*).  Views share it through an
/// `Rc`; when the last one goes the buffer returns to the thread's free list and the next owned result reuses it -- a model that
/// calls `split_owned` every forward keeps a constant number of device buffers (no per-call allocation, nothing leaked).
pub struct OwnedSlot(Slot);
impl OwnedSlot {
    pub fn slot(&self) -> Slot {
        self.0
    }
}
impl Drop for OwnedSlot {
    fn drop(&mut self) {
        // Not through the runtime: an owned view may be dropped while the runtime is borrowed (inside a kernel wrapper's with_rt
        // closure, or in a nested drop), and a slot that waited for that borrow would be lost.  The slot goes onto a list of its
        // own that pooled_slot() drains.  At thread exit the list may already be gone: the ctx then releases the buffer with
        // everything else.
        let _ = RETURNED.try_with(|cell| cell.borrow_mut().push(self.0));
    }
}
pub fn pooled_slot() -> Rc<OwnedSlot> {
    let back: Vec<Slot> = RETURNED.with(|cell| std::mem::take(&mut *cell.borrow_mut()));
    with_rt(|r| {
        r.free.extend(back);
        let s = match r.free.pop() {
            Some(s) => s,
            None => new_buf(r),
        };
        Rc::new(OwnedSlot(s))
    })
}
thread_local! {
    // buffers whose last owned view has gone, until the next pooled_slot() call moves them to the runtime's free list
    static RETURNED: RefCell<Vec<Slot>> = RefCell::new(Vec::new());
    // one ctx per host thread: lele itself is single-threaded with thread-local caches (conv2d.rs:601-603)
    static RT: RefCell<Option<Runtime>> = RefCell::new(None);
}
fn with_rt<R>(f: impl FnOnce(&mut Runtime) -> R) -> R {
    RT.with(|cell| {
        let mut g = cell.borrow_mut();
        if g.is_none() {
            let device = std::env::var("LELE_HIP_DEVICE").ok().and_then(|v| v.parse().ok()).unwrap_or(0);
            let mut ctx = std::ptr::null_mut();
            check(unsafe { ffi::lele_hip_ctx_create(device, &mut ctx) });
            *g = Some(Runtime { ctx, slots: HashMap::new(), scratch: Vec::new(), free: Vec::new(), prepared: HashMap::new() });
        }
        f(g.as_mut().unwrap())
    })
}
pub fn ctx() -> *mut ffi::LeleCtx {
    with_rt(|r| r.ctx)
}
fn new_buf(r: &mut Runtime) -> Slot {
    let mut b = std::ptr::null_mut();
    check(unsafe { ffi::lele_hip_buf_create(r.ctx, &mut b) });
    Slot(b)
}
pub fn slot_of<T>(out: &mut Vec<T>) -> Slot {
    let key = out as *mut Vec<T> as usize;
    with_rt(|r| {
        if let Some(s) = r.slots.get(&key) {
            return *s;
        }
        let s = new_buf(r);
        r.slots.insert(key, s);
        s
    })
}
pub fn scratch_slot(i: usize) -> Slot {
    with_rt(|r| {
        while r.scratch.len() <= i {
            let s = new_buf(r);
            r.scratch.push(s);
        }
        r.scratch[i]
    })
}
/// lele's kernels panic! on shape / attribute violations; the library reports them as a non-zero status + message
pub fn check(rc: c_int) {
    if rc != 0 {
        let msg = unsafe { CStr::from_ptr(ffi::lele_hip_last_error()) }.to_string_lossy().into_owned();
        panic!("{}", msg);
    }
}
pub fn sync() {
    check(unsafe { ffi::lele_hip_sync(ctx()) });
}
pub fn download<T: ElementOps>(slot: Slot, len: usize) -> Vec<T> {
    let mut v = vec![T::default(); len];
    if len > 0 {
        check(unsafe { ffi::lele_hip_buf_to_host(slot.raw(), v.as_mut_ptr() as *mut c_void, len * std::mem::size_of::<T>()) });
    }
    v
}
pub fn download_f32(slot: Slot, dst: &mut [f32]) {
    check(unsafe { ffi::lele_hip_buf_to_host(slot.raw(), dst.as_mut_ptr() as *mut c_void, dst.len() * 4) });
}
pub fn read_f32(slot: Slot) -> f32 {
    download::<f32>(slot, 1)[0]
}
pub fn opt_c<T: ElementOps>(t: Option<&TensorView<'_, T>>) -> OptC {
    OptC(t.map(|v| v.as_c()))
}
pub struct OptC(Option<crate::tensor::CView>);
impl OptC {
    pub fn ptr(&self) -> *const ffi::LeleTensor {
        self.0.as_ref().map(|c| c.ptr()).unwrap_or(std::ptr::null())
    }
}
pub fn scalar_opt<U: ElementOps>(t: Option<&TensorView<U>>) -> (c_int, f32) {
    match t {
        Some(v) if !v.data.is_empty() => (1, v.data[0].as_f32()),
        _ => (0, 0.0),
    }
}

/// result shape written by a kernel (out_shape / out_rank of the C ABI)
pub struct Shape {
    dims: [i64; ffi::LELE_MAX_RANK],
    rank: i32,
}
impl Shape {
    pub fn new() -> Self {
        Shape { dims: [0; ffi::LELE_MAX_RANK], rank: 0 }
    }
    pub fn dims(&mut self) -> *mut i64 {
        self.dims.as_mut_ptr()
    }
    pub fn rank(&mut self) -> *mut i32 {
        &mut self.rank
    }
    pub fn vec(&self) -> Vec<usize> {
        self.dims[..self.rank as usize].iter().map(|&d| d as usize).collect()
    }
}

// ---------------------------------------------------------------------------------------------- prepared weights
/// quantization.rs:198-215.  The handle owns the packed device copy; dropped with the thread's runtime.
pub struct PreparedWeights {
    h: *mut ffi::LelePrepared,
    pub k: usize,
    pub n: usize,
}
impl PreparedWeights {
    pub fn new(b_data: &[u8], k: usize, n: usize) -> Self {
        assert_eq!(b_data.len(), k * n);
        let mut h = std::ptr::null_mut();
        check(unsafe { ffi::lele_hip_prepare_weights(ctx(), b_data.as_ptr(), k as i64, n as i64, &mut h) });
        PreparedWeights { h, k, n }
    }
    pub fn raw(&self) -> *const ffi::LelePrepared {
        self.h
    }
}
impl Drop for PreparedWeights {
    fn drop(&mut self) {
        unsafe { ffi::lele_hip_prepared_destroy(self.h) };
    }
}
pub struct PreparedRef(*const ffi::LelePrepared);
impl PreparedRef {
    pub fn raw(&self) -> *const ffi::LelePrepared {
        self.0
    }
}
/// mat_mul_integer_u8_weights (quantization.rs:173) takes the raw weight slice on every call: one handle per (pointer, length)
pub fn prepared_for(b_u8: &[u8], b_shape: &[usize]) -> PreparedRef {
    let key = (b_u8.as_ptr() as usize, b_u8.len());
    let (k, n) = (b_shape[b_shape.len() - 2], b_shape[b_shape.len() - 1]);
    with_rt(|r| {
        if !r.prepared.contains_key(&key) {
            let mut h = std::ptr::null_mut();
            check(unsafe { ffi::lele_hip_prepare_weights(r.ctx, b_u8.as_ptr(), k as i64, n as i64, &mut h) });
            r.prepared.insert(key, PreparedWeights { h, k, n });
        }
        PreparedRef(r.prepared[&key].h)
    })
}

// ---------------------------------------------------------------------------------------------- views (shape.rs)
fn resolve(shape: &[usize], target: &[i64]) -> Vec<usize> {
    // shape.rs:2-52: 0 copies the input dimension, one -1 is inferred
    let total: usize = shape.iter().product();
    let mut out: Vec<usize> = Vec::with_capacity(target.len());
    let mut infer = None;
    for (i, &d) in target.iter().enumerate() {
        if d == -1 {
            infer = Some(i);
            out.push(1);
        } else if d == 0 {
            out.push(*shape.get(i).unwrap_or(&1));
        } else {
            out.push(d as usize);
        }
    }
    if let Some(i) = infer {
        let known: usize = out.iter().product();
        out[i] = if known == 0 { 0 } else { total / known };
    }
    out
}
pub fn reshape<'a, T: ElementOps>(input: &TensorView<'a, T>, target: &[i64]) -> TensorView<'a, T> {
    input.with_shape(resolve(&input.shape, target))
}
pub fn flatten<'a, T: ElementOps>(input: &TensorView<'a, T>, axis: i64) -> TensorView<'a, T> {
    let r = input.shape.len() as i64;
    let ax = (if axis < 0 { axis + r } else { axis }) as usize;
    let a: usize = input.shape[..ax].iter().product();
    let b: usize = input.shape[ax..].iter().product();
    input.with_shape(vec![a, b])
}
pub fn unsqueeze<'a, T: ElementOps>(input: &TensorView<'a, T>, axes: &[i64]) -> TensorView<'a, T> {
    let r = (input.shape.len() + axes.len()) as i64;
    let mut at: Vec<usize> = axes.iter().map(|&a| (if a < 0 { a + r } else { a }) as usize).collect();
    at.sort_unstable();
    let mut shape = input.shape.to_vec();
    for a in at {
        shape.insert(a, 1);
    }
    input.with_shape(shape)
}
pub fn squeeze<'a, T: ElementOps>(input: &TensorView<'a, T>, axes: Option<&[i64]>) -> TensorView<'a, T> {
    let r = input.shape.len() as i64;
    let shape: Vec<usize> = match axes {
        Some(ax) if !ax.is_empty() => {
            let drop: Vec<usize> = ax.iter().map(|&a| (if a < 0 { a + r } else { a }) as usize).collect();
            input.shape.iter().enumerate().filter(|(i, _)| !drop.contains(i)).map(|(_, &d)| d).collect()
        }
        _ => input.shape.iter().copied().filter(|&d| d != 1).collect(),
    };
    input.with_shape(shape)
}

// ---------------------------------------------------------------------------------------------- strided-copy geometry
pub struct Geom {
    pub dims: Vec<i64>,
    pub strides: Vec<i64>,
    pub mods: Vec<i64>,
    pub offset: i64,
}
fn row_major(shape: &[usize]) -> Vec<i64> {
    let mut s = vec![1i64; shape.len()];
    for i in (0..shape.len().saturating_sub(1)).rev() {
        s[i] = s[i + 1] * shape[i + 1] as i64;
    }
    s
}
/// manipulation.rs:209-380: per axis start / end clamped as ONNX Slice prescribes, negative indices from the end, any step
pub fn slice_geometry(shape: &[usize], starts: &[i64], ends: &[i64], axes: &[i64], steps: &[i64]) -> Geom {
    let r = shape.len();
    let st = row_major(shape);
    let (mut dims, mut strides, mut offset) = (shape.iter().map(|&d| d as i64).collect::<Vec<_>>(), st.clone(), 0i64);
    for i in 0..starts.len() {
        let ax = if axes.is_empty() { i as i64 } else { axes[i] };
        let ax = (if ax < 0 { ax + r as i64 } else { ax }) as usize;
        let d = shape[ax] as i64;
        let step = if steps.is_empty() { 1 } else { steps[i] };
        assert!(step != 0, "Slice: step must not be 0");
        let (mut s, mut e) = (starts[i], ends[i]);
        if s < 0 { s += d; }
        if e < 0 { e += d; }
        let n = if step > 0 {
            s = s.clamp(0, d);
            e = e.clamp(0, d);
            if e > s { (e - s + step - 1) / step } else { 0 }
        } else {
            s = s.clamp(-1, d - 1);
            e = e.clamp(-1, d - 1);
            if s > e { (s - e - step - 1) / (-step) } else { 0 }
        };
        offset += s.max(0) * st[ax];
        dims[ax] = n;
        strides[ax] = st[ax] * step;
    }
    Geom { dims, strides, mods: Vec::new(), offset }
}
pub fn transpose_geometry(shape: &[usize], perm: &[i64]) -> Geom {
    let r = shape.len() as i64;
    let st = row_major(shape);
    let p: Vec<usize> = if perm.is_empty() { (0..shape.len()).rev().collect() } else { perm.iter().map(|&a| (if a < 0 { a + r } else { a }) as usize).collect() };
    Geom { dims: p.iter().map(|&a| shape[a] as i64).collect(), strides: p.iter().map(|&a| st[a]).collect(), mods: Vec::new(), offset: 0 }
}
pub fn expand_geometry(shape: &[usize], target: &[i64]) -> Geom {
    // math.rs:2168-2247: numpy broadcasting of the input against `target`
    let r = shape.len().max(target.len());
    let st = row_major(shape);
    let (mut dims, mut strides) = (vec![1i64; r], vec![0i64; r]);
    for i in 0..r {
        let si = (i + shape.len()).checked_sub(r).map(|j| shape[j] as i64).unwrap_or(1);
        let ti = (i + target.len()).checked_sub(r).map(|j| target[j]).unwrap_or(1);
        assert!(si == ti || si == 1 || ti == 1, "Expand: incompatible shapes");
        dims[i] = si.max(ti);
        strides[i] = if si == 1 { 0 } else { st[i + shape.len() - r] };
    }
    Geom { dims, strides, mods: Vec::new(), offset: 0 }
}
pub fn tile_geometry(shape: &[usize], repeats: &[i64]) -> Geom {
    let st = row_major(shape);
    Geom { dims: shape.iter().zip(repeats).map(|(&d, &r)| d as i64 * r).collect(), strides: st, mods: shape.iter().map(|&d| d as i64).collect(), offset: 0 }
}
pub fn pad_args<T: ElementOps>(input: &TensorView<T>, pads: &[i64], constant_value: Option<&TensorView<T>>, mode: &str) -> (Vec<i64>, i32, u64) {
    // manipulation.rs:397-412: `pads` may cover only the trailing dimensions; negative entries are clamped to 0
    let r = input.shape.len();
    let half = pads.len() / 2;
    let mut full = vec![0i64; 2 * r];
    for i in 0..half {
        full[r - half + i] = pads[i].max(0);
        full[2 * r - half + i] = pads[half + i].max(0);
    }
    let mode_id = match mode { "constant" => 0, "edge" => 1, "reflect" => 2, m => panic!("Pad: unknown mode {}", m) };
    let mut fill = 0u64;
    if let Some(c) = constant_value {
        if let Some(v) = c.data.first() {
            let bytes = unsafe { std::slice::from_raw_parts(v as *const T as *const u8, std::mem::size_of::<T>()) };
            let mut raw = [0u8; 8];
            raw[..bytes.len()].copy_from_slice(bytes);
            fill = u64::from_le_bytes(raw);
        }
    }
    (full, mode_id, fill)
}
pub fn resize_target(shape: &[usize], scales: Option<&[f32]>, sizes: Option<&[i64]>) -> (i64, i64) {
    if let Some(s) = sizes {
        if s.len() == 4 {
            return (s[2], s[3]);
        }
    }
    let sc = scales.expect("Resize: neither sizes nor scales");
    ((shape[2] as f32 * sc[2]).floor() as i64, (shape[3] as f32 * sc[3]).floor() as i64)
}
pub fn split_into<'a, T: ElementOps>(input: &TensorView<'_, T>, axis: i64, splits: &[i64], outputs: &'a mut [Vec<T>]) -> Vec<TensorView<'a, T>> {
    let r = input.shape.len() as i64;
    let ax = (if axis < 0 { axis + r } else { axis }) as usize;
    let st = row_major(&input.shape);
    let mut start = 0i64;
    let mut views = Vec::with_capacity(splits.len());
    let c = input.as_c();
    for (out, &len) in outputs.iter_mut().zip(splits) {
        let mut dims: Vec<i64> = input.shape.iter().map(|&d| d as i64).collect();
        dims[ax] = len;
        let slot = slot_of(out);
        let mut sh = Shape::new();
        check(unsafe { ffi::lele_hip_strided_copy(ctx(), c.ptr(), dims.as_ptr(), st.as_ptr(), std::ptr::null(), dims.len() as i32, start * st[ax], slot.raw(), sh.dims(), sh.rank()) });
        views.push(TensorView::device(slot, sh.vec()));
        start += len;
    }
    views
}
pub fn split_owned<T: ElementOps>(input: &TensorView<'_, T>, axis: i64, splits: &[i64]) -> Vec<TensorView<'static, T>> {
    // manipulation.rs:1150-1213: owned results -- every part in a pooled buffer that its views keep alive and give back
    let r = input.shape.len() as i64;
    let ax = (if axis < 0 { axis + r } else { axis }) as usize;
    let st = row_major(&input.shape);
    let c = input.as_c();
    let mut start = 0i64;
    let mut views = Vec::with_capacity(splits.len());
    for &len in splits {
        let mut dims: Vec<i64> = input.shape.iter().map(|&d| d as i64).collect();
        dims[ax] = len;
        let keep = pooled_slot();
        let mut sh = Shape::new();
        check(unsafe { ffi::lele_hip_strided_copy(ctx(), c.ptr(), dims.as_ptr(), st.as_ptr(), std::ptr::null(), dims.len() as i32, start * st[ax], keep.slot().raw(), sh.dims(), sh.rank()) });
        views.push(TensorView::device_owned(keep, sh.vec()));
        start += len;
    }
    views
}
/// conv2d.rs:75,101 (`reset_conv_stats` / `print_conv_stats`, called by examples/yolo26n-seg/src/main.rs:64,74; no-ops upstream):
/// here they front the library's convolution counters (calls and multiply-accumulates issued on this thread's context)
pub fn reset_conv_stats() {
    check(unsafe { ffi::lele_hip_conv_stats_reset(ctx()) });
}
pub fn print_conv_stats() {
    let (mut calls, mut macs) = (0i64, 0i64);
    check(unsafe { ffi::lele_hip_conv_stats(ctx(), &mut calls, &mut macs) });
    println!("conv stats: {} convolution calls, {:.3} GMAC", calls, macs as f64 * 1e-9);
}
pub fn min_max(input: &TensorView<'_, f32>) -> (f32, f32) {
    input.data.iter().fold((f32::MAX, f32::MIN), |(a, b), &v| (a.min(v), b.max(v)))
}
/// the f32 0 / 1 flags a comparison left in `flags` (a scratch slot), as the i64 tensor lele's `*_i64` comparisons return
pub fn flags_to_i64<'a>(flags: Slot, shape: Vec<usize>, out: &'a mut Vec<i64>) -> TensorView<'a, i64> {
    cast_into(&TensorView::<f32>::device(flags, shape), out)
}
fn cast_into<'b, T: ElementOps, D: ElementOps>(input: &TensorView<'_, T>, out: &'b mut Vec<D>) -> TensorView<'b, D> {
    let slot = slot_of(out);
    let mut sh = Shape::new();
    check(unsafe { ffi::lele_hip_cast(ctx(), input.as_c().ptr(), D::DTYPE, slot.raw(), sh.dims(), sh.rank()) });
    TensorView::device(slot, sh.vec())
}
/// utils.rs:71-97: ONNX Cast to f32 / to i64 (`lele::kernels::utils::cast_to_f32` / `cast_to_i64`), one device pass, the result in `out`'s slot
pub fn cast_to_f32<'a, 'b, T: ElementOps>(input: &TensorView<'a, T>, out: &'b mut Vec<f32>) -> TensorView<'b, f32> {
    cast_into(input, out)
}
pub fn cast_to_i64<'a, 'b, T: ElementOps>(input: &TensorView<'a, T>, out: &'b mut Vec<i64>) -> TensorView<'b, i64> {
    cast_into(input, out)
}
pub fn constant_of_shape<'a, 'b, T: ElementOps + AsI64, V: ElementOps>(input: &TensorView<'a, T>, value: V, out: &'b mut Vec<V>) -> TensorView<'b, V> {
    let dims: Vec<i64> = input.data.iter().map(|v| v.as_i64()).collect(); // a shape tensor: a host read
    let bytes = unsafe { std::slice::from_raw_parts(&value as *const V as *const u8, std::mem::size_of::<V>()) };
    let mut raw = [0u8; 8];
    raw[..bytes.len()].copy_from_slice(bytes);
    let slot = slot_of(out);
    let mut sh = Shape::new();
    check(unsafe { ffi::lele_hip_fill(ctx(), dims.as_ptr(), dims.len() as i32, V::DTYPE, u64::from_le_bytes(raw), slot.raw(), sh.dims(), sh.rank()) });
    TensorView::device(slot, sh.vec())
}
