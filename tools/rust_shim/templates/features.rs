//! `lele::features::{FeatureConfig, SenseVoiceFrontend, Cmvn, Lfr}` (src/features/pipeline.rs:8-193, cmvn.rs, lfr.rs).
use crate::ffi;
use crate::rt::{self, Shape};
use crate::tensor::TensorView;

/// pipeline.rs:8-27
#[derive(Clone, Copy, Debug)]
pub struct FeatureConfig {
    pub sample_rate: usize,
    pub n_mels: usize,
    pub frame_length_ms: f32,
    pub frame_shift_ms: f32,
    pub lfr_m: usize,
    pub lfr_n: usize,
}
impl Default for FeatureConfig {
    fn default() -> Self {
        FeatureConfig { sample_rate: 16000, n_mels: 80, frame_length_ms: 25.0, frame_shift_ms: 10.0, lfr_m: 7, lfr_n: 6 }
    }
}

pub struct SenseVoiceFrontend {
    h: *mut ffi::LeleFrontend,
    out: Vec<f32>,
}
impl SenseVoiceFrontend {
    /// pipeline.rs:38-65
    pub fn new(config: FeatureConfig) -> Self {
        let c = ffi::LeleFeatureConfig { sample_rate: config.sample_rate as i64, n_mels: config.n_mels as i64, frame_length_ms: config.frame_length_ms,
                                         frame_shift_ms: config.frame_shift_ms, lfr_m: config.lfr_m as i64, lfr_n: config.lfr_n as i64 };
        let mut h = std::ptr::null_mut();
        rt::check(unsafe { ffi::lele_hip_frontend_create(rt::ctx(), &c, &mut h) });
        SenseVoiceFrontend { h, out: Vec::new() }
    }
    /// pipeline.rs:67-193: PCM -> [T, n_mels * lfr_m] (TensorView::empty() when the utterance is shorter than one frame)
    pub fn compute(&mut self, pcm: &[f32]) -> TensorView<'_, f32> {
        let x = TensorView::from_slice(pcm, vec![pcm.len()]);
        let slot = rt::slot_of(&mut self.out);
        let mut sh = Shape::new();
        rt::check(unsafe { ffi::lele_hip_frontend_compute(self.h, x.as_c().ptr(), slot.raw(), sh.dims(), sh.rank()) });
        if sh.vec().iter().product::<usize>() == 0 {
            return TensorView::empty();
        }
        TensorView::device(slot, sh.vec())
    }
    /// the same for `batch` equal-length utterances stored back to back ([batch, len] -> [batch, T, D]): one launch
    pub fn compute_batch<'a>(&'a mut self, pcm: &TensorView<'_, f32>) -> TensorView<'a, f32> {
        let slot = rt::slot_of(&mut self.out);
        let mut sh = Shape::new();
        rt::check(unsafe { ffi::lele_hip_frontend_compute_batch(self.h, pcm.as_c().ptr(), slot.raw(), sh.dims(), sh.rank()) });
        TensorView::device(slot, sh.vec())
    }
}
impl Drop for SenseVoiceFrontend {
    fn drop(&mut self) {
        unsafe { ffi::lele_hip_frontend_destroy(self.h) };
    }
}

/// cmvn.rs:14-92
pub struct Cmvn {
    pub eps: f32,
    out: Vec<f32>,
}
impl Cmvn {
    pub fn new() -> Self {
        Cmvn { eps: 1e-5, out: Vec::new() }
    }
    pub fn compute<'a>(&'a mut self, x: &TensorView<'_, f32>) -> TensorView<'a, f32> {
        let slot = rt::slot_of(&mut self.out);
        let mut sh = Shape::new();
        rt::check(unsafe { ffi::lele_hip_cmvn(rt::ctx(), x.as_c().ptr(), self.eps, slot.raw(), sh.dims(), sh.rank()) });
        TensorView::device(slot, sh.vec())
    }
    pub fn apply_with_stats<'a>(&'a mut self, x: &TensorView<'_, f32>, mean: &TensorView<'_, f32>, std: &TensorView<'_, f32>) -> TensorView<'a, f32> {
        let slot = rt::slot_of(&mut self.out);
        let mut sh = Shape::new();
        rt::check(unsafe { ffi::lele_hip_cmvn_apply_with_stats(rt::ctx(), x.as_c().ptr(), mean.as_c().ptr(), std.as_c().ptr(), self.eps, slot.raw(), sh.dims(), sh.rank()) });
        TensorView::device(slot, sh.vec())
    }
}

/// lfr.rs:18-54
pub struct Lfr {
    pub m: usize,
    pub n: usize,
    out: Vec<f32>,
}
impl Lfr {
    pub fn new(m: usize, n: usize) -> Self {
        Lfr { m, n, out: Vec::new() }
    }
    pub fn compute<'a>(&'a mut self, x: &TensorView<'_, f32>) -> TensorView<'a, f32> {
        let slot = rt::slot_of(&mut self.out);
        let mut sh = Shape::new();
        rt::check(unsafe { ffi::lele_hip_lfr(rt::ctx(), x.as_c().ptr(), self.m as i64, self.n as i64, slot.raw(), sh.dims(), sh.rank()) });
        TensorView::device(slot, sh.vec())
    }
}
