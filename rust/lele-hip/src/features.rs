// Hand-written part of the crate (tools/rust_shim/gen.py generates ffi.rs, kernels.rs and lib.rs beside it and leaves this file alone).
//! `lele::features::*` (src/features/mod.rs:1-12): the public interface of lele's audio front-end, same names, same signatures,
//! same receivers -- `SenseVoiceFrontend::new(config)` / `frontend.compute(&audio)` on an immutable binding, `Cmvn::default()`,
//! `Lfr::new(LfrConfig)`, and the building blocks (`hann_window`, `RealFft`, `SparseMelBank`, ...) lele's own tests call
//! (tests/verify_features.rs:6-64).  The data-parallel entry points run on the device and return device-resident, reference-counted
//! results (`TensorView<'static>` whose buffer goes back to the thread's pool when the last view of it is dropped); the small
//! table builders call the library's own (lele_hip_hann_window, lele_hip_mel_filterbank): this file holds no arithmetic of its own.
use crate::ffi;
use crate::rt::{self, Shape};
use crate::tensor::TensorView;

// ------------------------------------------------------------------------------------------------ window.rs
/// window.rs:2-12: the symmetric Hann window, as the library builds it for its own front-end (lele_hip_hann_window)
pub fn hann_window(size: usize) -> Vec<f32> {
    let mut w = vec![0.0f32; size];
    rt::check(unsafe { ffi::lele_hip_hann_window(size as i64, w.as_mut_ptr()) });
    w
}
/// window.rs:13-18
pub fn apply_window(input: &mut [f32], window: &[f32]) {
    assert_eq!(input.len(), window.len());
    for i in 0..input.len() {
        input[i] *= window[i];
    }
}

// ------------------------------------------------------------------------------------------------ mel.rs
/// mel.rs:1-3
pub fn hz_to_mel_htk(hz: f32) -> f32 {
    unsafe { ffi::lele_hip_hz_to_mel_htk(hz) }
}
/// mel.rs:4-6
pub fn mel_to_hz_htk(mel: f32) -> f32 {
    unsafe { ffi::lele_hip_mel_to_hz_htk(mel) }
}
/// mel.rs:7-45: dense HTK triangles, row-major [n_mels, n_fft / 2 + 1] (lele_hip_mel_filterbank: the table the device front-end uses)
pub fn mel_filterbank(sample_rate: f32, n_fft: usize, n_mels: usize, f_min: f32, f_max: Option<f32>) -> Vec<f32> {
    let mut bank = vec![0.0f32; n_mels * (n_fft / 2 + 1)];
    rt::check(unsafe {
        ffi::lele_hip_mel_filterbank(sample_rate, n_fft as i64, n_mels as i64, f_min, f_max.is_some() as i32, f_max.unwrap_or(0.0), bank.as_mut_ptr())
    });
    bank
}
/// mel.rs:47-104: every filter as (first non-zero bin, its run of weights)
#[derive(Clone, Debug)]
pub struct SparseMelBank {
    pub n_mels: usize,
    pub n_freqs: usize,
    pub filters: Vec<(usize, Vec<f32>)>,
}
impl SparseMelBank {
    pub fn new(sample_rate: f32, n_fft: usize, n_mels: usize, f_min: f32, f_max: Option<f32>) -> Self {
        let dense = mel_filterbank(sample_rate, n_fft, n_mels, f_min, f_max);
        let n_freqs = n_fft / 2 + 1;
        let filters = dense
            .chunks(n_freqs)
            .map(|row| match (row.iter().position(|&w| w != 0.0), row.iter().rposition(|&w| w != 0.0)) {
                (Some(a), Some(b)) => (a, row[a..=b].to_vec()),
                _ => (0, Vec::new()),
            })
            .collect();
        SparseMelBank { n_mels, n_freqs, filters }
    }
    pub fn apply(&self, power_spectrum: &[f32], output: &mut [f32]) {
        assert_eq!(power_spectrum.len(), self.n_freqs);
        assert_eq!(output.len(), self.n_mels);
        for (o, (start, w)) in output.iter_mut().zip(self.filters.iter()) {
            let mut acc = 0.0f32; // mel.rs:96-101: mul then add, in bin order
            for (k, wk) in w.iter().enumerate() {
                acc += wk * power_spectrum[start + k];
            }
            *o = acc;
        }
    }
}
/// mel.rs:106-123
pub fn apply_mel_bank(power_spectrum: &[f32], mel_filters: &[f32], n_mels: usize, output: &mut [f32]) {
    let bins = power_spectrum.len();
    assert_eq!(mel_filters.len(), n_mels * bins);
    assert_eq!(output.len(), n_mels);
    for (o, row) in output.iter_mut().zip(mel_filters.chunks(bins)) {
        let mut acc = 0.0f32;
        for j in 0..bins {
            acc += row[j] * power_spectrum[j];
        }
        *o = acc;
    }
}
/// mel.rs:124-128
pub fn log_compress(input: &mut [f32], eps: f32) {
    for x in input.iter_mut() {
        *x = x.max(eps).ln();
    }
}

// ------------------------------------------------------------------------------------------------ fft.rs
#[derive(Clone, Copy, Debug, Default)]
pub struct Complex<T> {
    pub re: T,
    pub im: T,
}
/// fft.rs:1-50: a real FFT of fixed power-of-two length; the transform itself runs on the device (`lele_hip_rfft`: lele's
/// radix-2 network bit for bit), bins n/2 + 1 .. n - 1 are the conjugate mirror as upstream fills them
pub struct RealFft {
    n: usize,
}
impl RealFft {
    pub fn new(length: usize) -> Self {
        RealFft { n: length }
    }
    pub fn scratch_len(&self) -> usize {
        self.n
    }
    pub fn process_with_scratch(&self, input: &[f32], output: &mut [Complex<f32>], _scratch: &mut [Complex<f32>]) {
        assert_eq!(input.len(), self.n);
        let x = TensorView::from_slice(input, vec![self.n]);
        let (re_slot, im_slot) = (rt::pooled_slot(), rt::pooled_slot());
        let mut sh = Shape::new();
        rt::check(unsafe { ffi::lele_hip_rfft(rt::ctx(), x.as_c().ptr(), re_slot.slot().raw(), im_slot.slot().raw(), sh.dims(), sh.rank()) });
        let half = self.n / 2 + 1;
        let (re, im) = (rt::download::<f32>(re_slot.slot(), half), rt::download::<f32>(im_slot.slot(), half));
        for i in 0..half {
            output[i] = Complex { re: re[i], im: im[i] };
        }
        for i in half..self.n {
            output[i] = Complex { re: re[self.n - i], im: -im[self.n - i] };
        }
    }
    pub fn process(&self, input: &[f32], output: &mut [Complex<f32>]) {
        let mut scratch = vec![Complex::default(); self.scratch_len()];
        self.process_with_scratch(input, output, &mut scratch);
    }
}

// ------------------------------------------------------------------------------------------------ lfr.rs
pub struct LfrConfig {
    pub m: usize,
    pub n: usize,
}
impl Default for LfrConfig {
    fn default() -> Self {
        LfrConfig { m: 7, n: 6 }
    }
}
pub struct Lfr {
    config: LfrConfig,
}
impl Lfr {
    pub fn new(config: LfrConfig) -> Self {
        Lfr { config }
    }
    /// lfr.rs:18-54: [T, D] or [1, T, D] -> [ceil(T / n), D * m]
    pub fn compute(&self, input: &TensorView) -> TensorView<'static> {
        let keep = rt::pooled_slot();
        let mut sh = Shape::new();
        rt::check(unsafe { ffi::lele_hip_lfr(rt::ctx(), input.as_c().ptr(), self.config.m as i64, self.config.n as i64, keep.slot().raw(), sh.dims(), sh.rank()) });
        TensorView::device_owned(keep, sh.vec())
    }
}

// ------------------------------------------------------------------------------------------------ cmvn.rs
pub struct Cmvn {
    eps: f32,
}
impl Default for Cmvn {
    fn default() -> Self {
        Cmvn { eps: 1e-5 }
    }
}
impl Cmvn {
    pub fn new(eps: f32) -> Self {
        Cmvn { eps }
    }
    /// cmvn.rs:14-66
    pub fn compute(&self, input: &TensorView) -> TensorView<'static> {
        let keep = rt::pooled_slot();
        let mut sh = Shape::new();
        rt::check(unsafe { ffi::lele_hip_cmvn(rt::ctx(), input.as_c().ptr(), self.eps, keep.slot().raw(), sh.dims(), sh.rank()) });
        TensorView::device_owned(keep, sh.vec())
    }
    /// cmvn.rs:67-92
    pub fn apply_with_stats(&self, input: &TensorView, mean: &[f32], std: &[f32]) -> TensorView<'static> {
        let (m, s) = (TensorView::from_slice(mean, vec![mean.len()]), TensorView::from_slice(std, vec![std.len()]));
        let keep = rt::pooled_slot();
        let mut sh = Shape::new();
        rt::check(unsafe {
            ffi::lele_hip_cmvn_apply_with_stats(rt::ctx(), input.as_c().ptr(), m.as_c().ptr(), s.as_c().ptr(), self.eps, keep.slot().raw(), sh.dims(), sh.rank())
        });
        TensorView::device_owned(keep, sh.vec())
    }
}

// ------------------------------------------------------------------------------------------------ pipeline.rs
/// pipeline.rs:8-27
#[derive(Debug, Clone)]
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
}
impl SenseVoiceFrontend {
    /// pipeline.rs:38-65
    pub fn new(config: FeatureConfig) -> Self {
        let c = ffi::LeleFeatureConfig {
            sample_rate: config.sample_rate as i64,
            n_mels: config.n_mels as i64,
            frame_length_ms: config.frame_length_ms,
            frame_shift_ms: config.frame_shift_ms,
            lfr_m: config.lfr_m as i64,
            lfr_n: config.lfr_n as i64,
        };
        let mut h = std::ptr::null_mut();
        rt::check(unsafe { ffi::lele_hip_frontend_create(rt::ctx(), &c, &mut h) });
        SenseVoiceFrontend { h }
    }
    /// pipeline.rs:66-193: PCM -> [T, n_mels * lfr_m] (TensorView::empty() when the utterance is shorter than one frame)
    pub fn compute(&self, pcm: &[f32]) -> TensorView<'static> {
        let x = TensorView::from_slice(pcm, vec![pcm.len()]);
        let keep = rt::pooled_slot();
        let mut sh = Shape::new();
        rt::check(unsafe { ffi::lele_hip_frontend_compute(self.h, x.as_c().ptr(), keep.slot().raw(), sh.dims(), sh.rank()) });
        if sh.vec().iter().product::<usize>() == 0 {
            return TensorView::empty();
        }
        TensorView::device_owned(keep, sh.vec())
    }
    /// beyond upstream: `batch` equal-length utterances stored back to back ([batch, len] -> [batch, T, D]) in one launch
    pub fn compute_batch(&self, pcm: &TensorView) -> TensorView<'static> {
        let keep = rt::pooled_slot();
        let mut sh = Shape::new();
        rt::check(unsafe { ffi::lele_hip_frontend_compute_batch(self.h, pcm.as_c().ptr(), keep.slot().raw(), sh.dims(), sh.rank()) });
        TensorView::device_owned(keep, sh.vec())
    }
}
impl Drop for SenseVoiceFrontend {
    fn drop(&mut self) {
        unsafe { ffi::lele_hip_frontend_destroy(self.h) };
    }
}

// The paths lele's own sources and tests use (src/features/mod.rs:1-12 declares six submodules and re-exports their contents;
// tests/verify_features.rs:2 imports `lele::features::fft::Complex`): the same items under the same submodule names.
pub mod cmvn {
    pub use super::Cmvn;
}
pub mod fft {
    pub use super::{Complex, RealFft};
}
pub mod lfr {
    pub use super::{Lfr, LfrConfig};
}
pub mod mel {
    pub use super::{apply_mel_bank, hz_to_mel_htk, log_compress, mel_filterbank, mel_to_hz_htk, SparseMelBank};
}
pub mod pipeline {
    pub use super::{FeatureConfig, SenseVoiceFrontend};
}
pub mod window {
    pub use super::{apply_window, hann_window};
}
