// Hand-written part of the crate (tools/rust_shim/gen.py generates ffi.rs, kernels.rs and lib.rs beside it and leaves this file alone).
//! `lele::tensor::TensorView` (src/tensor.rs:5-166) with a device-aware payload.
//!
//! Upstream: `pub struct TensorView<'a, T = f32> { pub data: Cow<'a, [T]>, pub shape: Cow<'a, [usize]> }`.  Generated model code
//! only ever (a) passes views to kernels, (b) reads `.shape`, (c) reads `.data` of CONTROL values (TopK k, Resize sizes, If
//! predicates: ops/tensor.rs:567, ops/nn.rs:434-441, ops/control_flow.rs:43,49) and (d) calls the constructors / `to_owned`.
//! Here `data` is a `Payload` that derefs to `[T]`: host data is the upstream Cow; a kernel result lives in a `LeleBuf` and is
//! copied to the host the first time it is dereferenced (lazy D2H, cached) -- so chains of kernel calls never leave HBM and (c)
//! keeps working unchanged.
use crate::ffi;
use crate::rt::{self, ElementOps, OwnedSlot, Slot};
use std::borrow::Cow;
use std::cell::OnceCell;
use std::marker::PhantomData;
use std::ops::Deref;
use std::rc::Rc;

pub enum Payload<'a, T: Clone> {
    /// host memory (weights.bin slices, user inputs): staged / cached by the library per `mem`
    Host { data: Cow<'a, [T]>, weight: bool },
    /// a kernel result in the workspace slot `slot`; `host` is filled on first dereference.  `keep` is None for a slot that
    /// belongs to the caller's `out` Vec (the borrow `'a` keeps it alive, as upstream) and Some for an OWNED result
    /// (`split_owned`, `lele::features::*`): the buffer returns to the thread's pool when the last view of it is dropped
    Device { slot: Slot, len: usize, host: OnceCell<Vec<T>>, keep: Option<Rc<OwnedSlot>>, _borrow: PhantomData<&'a mut Vec<T>> },
}

impl<'a, T: ElementOps> Deref for Payload<'a, T> {
    type Target = [T];
    fn deref(&self) -> &[T] {
        match self {
            Payload::Host { data, .. } => data,
            Payload::Device { slot, len, host, .. } => host.get_or_init(|| rt::download::<T>(*slot, *len)),
        }
    }
}

pub struct TensorView<'a, T: Clone = f32> {
    pub data: Payload<'a, T>,
    pub shape: Cow<'a, [usize]>,
}

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

impl<'a, T: ElementOps> TensorView<'a, T> {
    // ---- constructors, as src/tensor.rs:27-71
    pub fn new(data: Cow<'a, [T]>, shape: Cow<'a, [usize]>) -> Self {
        assert_eq!(data.len(), shape.iter().product::<usize>(), "Data length mismatch"); // tensor.rs:29
        Self { data: Payload::Host { data, weight: false }, shape }
    }
    pub fn from_owned(data: Vec<T>, shape: Vec<usize>) -> TensorView<'static, T> {
        TensorView::new(Cow::Owned(data), Cow::Owned(shape))
    }
    pub fn from_slice(data: &'a [T], shape: Vec<usize>) -> Self {
        Self::new(Cow::Borrowed(data), Cow::Owned(shape))
    }
    pub fn empty() -> TensorView<'static, T> {
        TensorView { data: Payload::Host { data: Cow::Owned(Vec::new()), weight: false }, shape: Cow::Owned(Vec::new()) }
    }
    /// a weights.bin slice: immutable for the life of the thread's ctx -> uploaded / pre-packed once (LELE_MEM_WEIGHT)
    pub fn weight(data: &'a [T], shape: &'a [usize]) -> Self {
        Self { data: Payload::Host { data: Cow::Borrowed(data), weight: true }, shape: Cow::Borrowed(shape) }
    }
    /// a kernel result living in `slot` (rt::slot_of(out)); borrows the caller's `out` Vec like upstream's `from_slice(out, ..)`
    pub fn device(slot: Slot, shape: Vec<usize>) -> Self {
        let len = shape.iter().product();
        Self { data: Payload::Device { slot, len, host: OnceCell::new(), keep: None, _borrow: PhantomData }, shape: Cow::Owned(shape) }
    }
    /// a kernel result in a pooled buffer this view (and every view made from it) keeps alive: upstream's owned `'static` results
    pub fn device_owned(keep: Rc<OwnedSlot>, shape: Vec<usize>) -> TensorView<'static, T> {
        let len = shape.iter().product();
        TensorView { data: Payload::Device { slot: keep.slot(), len, host: OnceCell::new(), keep: Some(keep), _borrow: PhantomData }, shape: Cow::Owned(shape) }
    }
    pub fn dim(&self) -> usize {
        self.shape.len()
    }
    pub fn size(&self, dim: usize) -> usize {
        self.shape[dim]
    }
    /// tensor.rs:73-85: an owned copy (forward() returns these: src/compiler/mod.rs:1291-1303) -- the one place data comes home
    pub fn to_owned(&self) -> TensorView<'static, T> {
        TensorView::from_owned(self.data.to_vec(), self.shape.to_vec())
    }
    pub fn detach(&self) -> TensorView<'static, T> {
        self.to_owned()
    }
    /// the same storage under the same shape (identity / cast, shape.rs): no copy for device payloads
    pub fn share(&self) -> TensorView<'a, T> {
        self.with_shape(self.shape.to_vec())
    }
    /// the same storage under another shape of equal size (reshape / flatten / squeeze / unsqueeze, shape.rs:2-121)
    pub fn with_shape(&self, shape: Vec<usize>) -> TensorView<'a, T> {
        assert_eq!(shape.iter().product::<usize>(), self.shape.iter().product::<usize>(), "Reshape: element count mismatch");
        let data = match &self.data {
            // a borrowed slice outlives the new view (`'a`): same pointer, same `weight` identity -- the library caches packed
            // weights by (pointer, bytes), so a weight must never be re-labelled onto a temporary copy
            Payload::Host { data: Cow::Borrowed(b), weight } => Payload::Host { data: Cow::Borrowed(*b), weight: *weight },
            // owned host data: the view gets its own copy, which is NOT a declared-immutable weight (its address dies with it)
            Payload::Host { data: Cow::Owned(v), .. } => Payload::Host { data: Cow::Owned(v.clone()), weight: false },
            Payload::Device { slot, len, keep, .. } => Payload::Device { slot: *slot, len: *len, host: OnceCell::new(), keep: keep.clone(), _borrow: PhantomData },
        };
        TensorView { data, shape: Cow::Owned(shape) }
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

// from_bytes_* (tensor.rs:87-166): weights.bin decoding; u8 / i8 tensors are carried as f32 values, exactly as upstream
impl TensorView<'static, f32> {
    pub fn from_bytes_u8(bytes: &[u8], shape: Vec<usize>) -> Self {
        TensorView::from_owned(bytes.iter().map(|&b| b as f32).collect(), shape)
    }
    pub fn from_bytes_i8(bytes: &[u8], shape: Vec<usize>) -> Self {
        TensorView::from_owned(bytes.iter().map(|&b| b as i8 as f32).collect(), shape)
    }
}
