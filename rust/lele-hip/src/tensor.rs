// Hand-written part of the crate (tools/rust_shim/gen.py generates ffi.rs, kernels.rs and lib.rs beside it and leaves this file alone).
//! `lele::tensor` (src/tensor.rs) with a device-aware payload: every public item of upstream's module -- `TensorView` and its
//! constructors / accessors (`:27-85`), `reinterpret_as_u8` (`:88-98`), `IntoLogits` (`:100-128`), the nine `from_bytes_*`
//! weights.bin decoders (`:131-257`), the seven `TensorView*` aliases (`:14-20`) and the `f16` / `bf16` re-export (`:1`) -- under
//! upstream's names, parameter types and return types (tests/test_rust_shim.py compares the declaration text with
//! tools/rust_shim/signatures.json["tensor"]), so that a lele-generated model source (`examples/yolo26n-seg/src/yolo26seg.rs`,
//! helper block `src/compiler/mod.rs:1135-1233`) compiles against this crate unchanged.
//!
//! Upstream: `pub struct TensorView<'a, T = f32> { pub data: Cow<'a, [T]>, pub shape: Cow<'a, [usize]> }`.  Generated model code
//! only ever (a) passes views to kernels, (b) reads `.shape`, (c) reads `.data` of CONTROL values (TopK k, Resize sizes, If
//! predicates: ops/tensor.rs:567, ops/nn.rs:434-441, ops/control_flow.rs:43,49; `Some(&w.data)` as `Option<&[f32]>`,
//! yolo26seg.rs:393) and (d) calls the constructors / `to_owned` / `clone`.
//! Here `data` is a `Payload` that derefs to `[T]`: host data is the upstream Cow; a kernel result lives in a `LeleBuf` and is
//! copied to the host the first time it is dereferenced (lazy D2H, cached) -- so chains of kernel calls never leave HBM and (c)
//! keeps working unchanged.
//!
//! Weights.  A `weights.bin` slice reaches the kernels through `self.weight_f32(off, len, &shape)` = `from_bytes_f32` (and the
//! u8 / i8 / f16 decoders): those constructors are where a host pointer becomes `LELE_MEM_WEIGHT` -- immutable for the life of
//! the thread's context, uploaded (and pre-packed where the kernel wants it) ONCE by the library, keyed by (pointer, bytes).
//! `from_bytes_f32` borrows the bytes in place when they are 4-byte aligned (weights.bin tensors are 16-byte aligned,
//! src/compiler/mod.rs:1381-1505), so the key is the mapped file's own address; the decoders that change the element type keep
//! their decoded f32 image in a per-thread table so that the address stays the same from call to call.
use crate::ffi;
use crate::rt::{self, ElementOps, OwnedSlot, Slot};
pub use half::{bf16, f16};
use std::borrow::Cow;
use std::cell::{OnceCell, RefCell};
use std::collections::HashMap;
use std::marker::PhantomData;
use std::ops::Deref;
use std::rc::Rc;

pub enum Payload<'a, T: Clone> {
    /// host memory (weights.bin slices, user inputs): staged / cached by the library per `mem`
    Host { data: Cow<'a, [T]>, weight: bool },
    /// a kernel result in the workspace slot `slot`; `host` is filled on first dereference through `fetch` (the element type's
    /// download routine, fixed where the view is made -- so dereferencing needs no bound beyond upstream's `T: Clone`).  `keep`
    /// is None for a slot that belongs to the caller's `out` Vec (the borrow `'a` keeps it alive, as upstream) and Some for an
    /// OWNED result (`split_owned`, `lele::features::*`): the buffer returns to the thread's pool when the last view of it is dropped
    Device { slot: Slot, len: usize, host: OnceCell<Vec<T>>, fetch: fn(Slot, usize) -> Vec<T>, keep: Option<Rc<OwnedSlot>>, _borrow: PhantomData<&'a mut Vec<T>> },
}

impl<'a, T: Clone> Deref for Payload<'a, T> {
    type Target = [T];
    fn deref(&self) -> &[T] {
        match self {
            Payload::Host { data, .. } => data,
            Payload::Device { slot, len, host, fetch, .. } => host.get_or_init(|| fetch(*slot, *len)),
        }
    }
}

/// the same storage once more: a borrowed slice stays the same pointer with the same `weight` identity (the library caches packed
/// weights by (pointer, bytes), so a weight must never be re-labelled onto a temporary copy); owned host data is copied and the
/// copy is NOT a declared-immutable weight (its address dies with it); a device result shares its slot
impl<'a, T: Clone> Clone for Payload<'a, T> {
    fn clone(&self) -> Self {
        match self {
            Payload::Host { data: Cow::Borrowed(b), weight } => Payload::Host { data: Cow::Borrowed(*b), weight: *weight },
            Payload::Host { data: Cow::Owned(v), .. } => Payload::Host { data: Cow::Owned(v.clone()), weight: false },
            Payload::Device { slot, len, host, fetch, keep, .. } => {
                Payload::Device { slot: *slot, len: *len, host: host.clone(), fetch: *fetch, keep: keep.clone(), _borrow: PhantomData }
            }
        }
    }
}

impl<'a, T: Clone> std::fmt::Debug for Payload<'a, T> {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        match self {
            Payload::Host { data, weight } => write!(f, "Host {{ len: {}, weight: {} }}", data.len(), weight),
            Payload::Device { len, host, .. } => write!(f, "Device {{ len: {}, on_host: {} }}", len, host.get().is_some()),
        }
    }
}

#[derive(Debug, Clone)]
pub struct TensorView<'a, T: Clone = f32> {
    pub data: Payload<'a, T>,
    pub shape: Cow<'a, [usize]>,
}

// src/tensor.rs:14-20
pub type TensorViewF32<'a> = TensorView<'a, f32>;
pub type TensorViewI8<'a> = TensorView<'a, i8>;
pub type TensorViewU8<'a> = TensorView<'a, u8>;
pub type TensorViewI32<'a> = TensorView<'a, i32>;
pub type TensorViewI64<'a> = TensorView<'a, i64>;
pub type TensorViewF16<'a> = TensorView<'a, f16>;
pub type TensorViewBF16<'a> = TensorView<'a, bf16>;

/// what the C ABI sees: a LeleTensor plus the i64 shape it points to
pub struct CView {
    shape: Vec<i64>,
    t: ffi::LeleTensor,
}
impl CView {
    pub fn ptr(&self) -> *const ffi::LeleTensor {
        &self.t
    }
}

fn numel(shape: &[usize]) -> usize {
    shape.iter().product()
}

// ---- upstream's interface (src/tensor.rs:22-85): same bounds, same parameter and return types
impl<'a, T> TensorView<'a, T>
where
    T: Clone + std::fmt::Debug,
    [T]: ToOwned<Owned = Vec<T>>,
{
    pub fn new(data: &'a [T], shape: &'a [usize]) -> Self {
        assert_eq!(data.len(), numel(shape), "Data length mismatch"); // tensor.rs:29
        Self { data: Payload::Host { data: Cow::Borrowed(data), weight: false }, shape: Cow::Borrowed(shape) }
    }

    pub fn from_owned(data: Vec<T>, shape: Vec<usize>) -> Self {
        assert_eq!(data.len(), numel(&shape), "Data length mismatch");
        Self { data: Payload::Host { data: Cow::Owned(data), weight: false }, shape: Cow::Owned(shape) }
    }

    /// the one place data comes home: forward() returns these (src/compiler/mod.rs:1291-1303)
    pub fn to_owned(&self) -> TensorView<'static, T> {
        TensorView::from_owned(self.data.to_vec(), self.shape.to_vec())
    }

    pub fn empty() -> Self {
        Self { data: Payload::Host { data: Cow::Borrowed(&[]), weight: false }, shape: Cow::Borrowed(&[]) }
    }

    pub fn dim(&self) -> usize {
        self.shape.len()
    }

    pub fn size(&self, dim: usize) -> usize {
        self.shape[dim]
    }

    pub fn from_slice(data: &'a [T], shape: Vec<usize>) -> Self {
        assert_eq!(data.len(), numel(&shape), "Data length mismatch");
        Self { data: Payload::Host { data: Cow::Borrowed(data), weight: false }, shape: Cow::Owned(shape) }
    }

    /// # Safety
    /// As upstream (tensor.rs:73-85): the result's lifetime is chosen by the caller and may outlive what it points to.  Host data
    /// is re-borrowed in place (same pointer, same `weight` identity); a device result names the same slot -- the slot outlives
    /// every view (it belongs to the thread's runtime), what the caller vouches for is that it is not REWRITTEN while `'b` lasts.
    pub unsafe fn detach<'b>(&self) -> TensorView<'b, T> {
        let data = match &self.data {
            Payload::Host { data, weight } => {
                Payload::Host { data: Cow::Borrowed(unsafe { std::slice::from_raw_parts(data.as_ptr(), data.len()) }), weight: *weight }
            }
            Payload::Device { slot, len, host, fetch, keep, .. } => {
                Payload::Device { slot: *slot, len: *len, host: host.clone(), fetch: *fetch, keep: keep.clone(), _borrow: PhantomData }
            }
        };
        TensorView { data, shape: Cow::Borrowed(unsafe { std::slice::from_raw_parts(self.shape.as_ptr(), self.shape.len()) }) }
    }
}

impl<'a> TensorView<'a, f32> {
    /// # Safety
    /// tensor.rs:88-98: the f32 values are u8 codes carried as f32 (dynamic_quantize_linear's y).  A device result is converted on
    /// the device (Rust's saturating `as u8`) into a pooled buffer; host data on the host.
    pub unsafe fn reinterpret_as_u8(&self) -> TensorView<'a, u8> {
        if let Payload::Device { .. } = &self.data {
            let keep = rt::pooled_slot();
            let mut sh = rt::Shape::new();
            rt::check(unsafe { ffi::lele_hip_cast(rt::ctx(), self.as_c().ptr(), ffi::LELE_U8, keep.slot().raw(), sh.dims(), sh.rank()) });
            return TensorView::device_owned(keep, sh.vec());
        }
        TensorView::from_owned(self.data.iter().map(|&x| x as u8).collect(), self.shape.to_vec())
    }
}

// ---- tensor.rs:100-128: what a model's forward() hands the decoder -- the logits alone or (logits, anything)
pub trait IntoLogits<'a, T>
where
    T: Clone + std::fmt::Debug,
    [T]: ToOwned<Owned = Vec<T>>,
{
    fn into_logits(self) -> TensorView<'a, T>;
}

impl<'a, T> IntoLogits<'a, T> for TensorView<'a, T>
where
    T: Clone + std::fmt::Debug,
    [T]: ToOwned<Owned = Vec<T>>,
{
    fn into_logits(self) -> TensorView<'a, T> {
        self
    }
}

impl<'a, T, U> IntoLogits<'a, T> for (TensorView<'a, T>, U)
where
    T: Clone + std::fmt::Debug,
    [T]: ToOwned<Owned = Vec<T>>,
{
    fn into_logits(self) -> TensorView<'a, T> {
        self.0
    }
}

// ---- the device side of a view (nothing upstream has a counterpart for)
impl<'a, T: ElementOps> TensorView<'a, T> {
    /// a weights.bin slice: immutable for the life of the thread's ctx -> uploaded / pre-packed once (LELE_MEM_WEIGHT)
    pub fn weight(data: &'a [T], shape: &'a [usize]) -> Self {
        assert_eq!(data.len(), numel(shape), "Data length mismatch");
        Self { data: Payload::Host { data: Cow::Borrowed(data), weight: true }, shape: Cow::Borrowed(shape) }
    }
    /// a kernel result living in `slot` (rt::slot_of(out)); borrows the caller's `out` Vec like upstream's `from_slice(out, ..)`
    pub fn device(slot: Slot, shape: Vec<usize>) -> Self {
        let len = numel(&shape);
        Self { data: Payload::Device { slot, len, host: OnceCell::new(), fetch: rt::download::<T>, keep: None, _borrow: PhantomData }, shape: Cow::Owned(shape) }
    }
    /// a kernel result in a pooled buffer this view (and every view made from it) keeps alive: upstream's owned `'static` results
    pub fn device_owned(keep: Rc<OwnedSlot>, shape: Vec<usize>) -> TensorView<'static, T> {
        let len = numel(&shape);
        TensorView { data: Payload::Device { slot: keep.slot(), len, host: OnceCell::new(), fetch: rt::download::<T>, keep: Some(keep), _borrow: PhantomData }, shape: Cow::Owned(shape) }
    }
    /// the same storage under the same shape (identity / cast, shape.rs): no copy for device payloads
    pub fn share(&self) -> TensorView<'a, T> {
        self.with_shape(self.shape.to_vec())
    }
    /// the same storage under another shape of equal size (reshape / flatten / squeeze / unsqueeze, shape.rs:2-121)
    pub fn with_shape(&self, shape: Vec<usize>) -> TensorView<'a, T> {
        assert_eq!(numel(&shape), numel(&self.shape), "Reshape: element count mismatch");
        TensorView { data: self.data.clone(), shape: Cow::Owned(shape) }
    }
    pub fn as_c(&self) -> CView {
        let shape: Vec<i64> = self.shape.iter().map(|&d| d as i64).collect();
        let (data, mem) = match &self.data {
            Payload::Host { data, weight } => (data.as_ptr() as *const std::os::raw::c_void, if *weight { ffi::LELE_MEM_WEIGHT } else { ffi::LELE_MEM_HOST }),
            Payload::Device { slot, .. } => (slot.data(), ffi::LELE_MEM_DEVICE),
        };
        let mut v = CView { shape, t: ffi::LeleTensor { data, shape: std::ptr::null(), rank: 0, dtype: T::DTYPE, mem } };
        v.t.shape = v.shape.as_ptr();
        v.t.rank = v.shape.len() as i32;
        v
    }
}

// ---------------------------------------------------------------------------------------------- weights.bin decoders
/// little-endian words of `W` bytes (a trailing partial word is ignored, as `chunks_exact` does upstream)
fn le_words<const W: usize>(bytes: &[u8]) -> impl Iterator<Item = [u8; W]> + '_ {
    bytes.chunks_exact(W).map(|c| {
        let mut w = [0u8; W];
        w.copy_from_slice(c);
        w
    })
}

thread_local! {
    // decoded f32 images of u8 / i8 / f16 weight slices, by (address, bytes, source element kind).  A boxed slice never moves or
    // shrinks, so its address is the stable key the library's weight cache (LELE_MEM_WEIGHT) needs; entries live as long as the
    // thread, like upstream's own per-thread tables (tensor.rs:153-155, 172-174).
    static DECODED: RefCell<HashMap<(usize, usize, u8), Box<[f32]>>> = RefCell::new(HashMap::new());
}

fn decoded_weight(bytes: &[u8], kind: u8, shape: Vec<usize>, decode: impl FnOnce(&[u8]) -> Vec<f32>) -> TensorView<'static, f32> {
    let key = (bytes.as_ptr() as usize, bytes.len(), kind);
    let (ptr, len) = DECODED.with(|table| {
        let mut table = table.borrow_mut();
        let image = table.entry(key).or_insert_with(|| decode(bytes).into_boxed_slice());
        (image.as_ptr(), image.len())
    });
    assert_eq!(len, numel(&shape), "Data length mismatch");
    let data: &'static [f32] = unsafe { std::slice::from_raw_parts(ptr, len) };
    TensorView { data: Payload::Host { data: Cow::Borrowed(data), weight: true }, shape: Cow::Owned(shape) }
}

impl<'a> TensorView<'a, f32> {
    /// tensor.rs:131-147 (`weight_f32`, src/compiler/mod.rs:1135): the bytes in place when 4-byte aligned -- and then a declared
    /// weight: every convolution / matmul weight of a generated model passes through here -- else an owned little-endian decode
    pub fn from_bytes_f32(bytes: &'a [u8], shape: &'a [usize]) -> Self {
        if bytes.as_ptr() as usize % std::mem::align_of::<f32>() == 0 {
            let words = unsafe { std::slice::from_raw_parts(bytes.as_ptr() as *const f32, bytes.len() / 4) };
            return TensorView::weight(words, shape);
        }
        let owned: TensorView<'static, f32> = TensorView::from_owned(le_words::<4>(bytes).map(f32::from_le_bytes).collect(), shape.to_vec());
        owned
    }

    /// tensor.rs:149-166 (`weight_u8`): u8 codes carried as f32, one decoded image per weight slice
    pub fn from_bytes_u8(bytes: &[u8], shape: Vec<usize>) -> TensorView<'static, f32> {
        decoded_weight(bytes, 0, shape, |b| b.iter().map(|&x| x as f32).collect())
    }

    /// tensor.rs:168-185 (`weight_i8`)
    pub fn from_bytes_i8(bytes: &[u8], shape: Vec<usize>) -> TensorView<'static, f32> {
        decoded_weight(bytes, 1, shape, |b| b.iter().map(|&x| x as i8 as f32).collect())
    }

    /// tensor.rs:187-196 (`weight_f16`): IEEE half -> f32, exact
    pub fn from_bytes_f16(bytes: &[u8], shape: Vec<usize>) -> TensorView<'static, f32> {
        decoded_weight(bytes, 2, shape, |b| le_words::<2>(b).map(|w| f16::from_bits(u16::from_le_bytes(w)).to_f32()).collect())
    }

    /// tensor.rs:198-209 (`weight_i64_f32`): control values (shapes, axes) -- owned host data, read on the host
    pub fn from_bytes_i64_as_f32(bytes: &[u8], shape: Vec<usize>) -> TensorView<'static, f32> {
        TensorView::from_owned(le_words::<8>(bytes).map(|w| i64::from_le_bytes(w) as f32).collect(), shape)
    }

    /// tensor.rs:211-220 (`weight_i32_f32`)
    pub fn from_bytes_i32_as_f32(bytes: &[u8], shape: Vec<usize>) -> TensorView<'static, f32> {
        TensorView::from_owned(le_words::<4>(bytes).map(|w| i32::from_le_bytes(w) as f32).collect(), shape)
    }
}

impl<'a> TensorView<'a, i64> {
    /// tensor.rs:224-234 (`weight_i64`)
    pub fn from_bytes_i64(bytes: &[u8], shape: Vec<usize>) -> TensorView<'static, i64> {
        TensorView::from_owned(le_words::<8>(bytes).map(i64::from_le_bytes).collect(), shape)
    }

    /// tensor.rs:236-244 (`weight_i32_i64`)
    pub fn from_bytes_i32_as_i64(bytes: &[u8], shape: Vec<usize>) -> TensorView<'static, i64> {
        TensorView::from_owned(le_words::<4>(bytes).map(|w| i32::from_le_bytes(w) as i64).collect(), shape)
    }
}

impl<'a> TensorView<'a, i32> {
    /// tensor.rs:248-256 (`weight_i32`)
    pub fn from_bytes_i32(bytes: &[u8], shape: Vec<usize>) -> TensorView<'static, i32> {
        TensorView::from_owned(le_words::<4>(bytes).map(i32::from_le_bytes).collect(), shape)
    }
}
