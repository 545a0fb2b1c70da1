//! `src/gpu.rs` for bliss-rs (`bliss-audio` 0.13.0): the per-song analysis hot path and the feature-vector distances on
//! an MI355X through `libblissgpu.so` (C ABI: `include/blissgpu.h`).  Enabled by a `gpu` cargo feature; the CPU path stays
//! behind `#[cfg(not(feature = "gpu"))]`.
//!
//! **UNTESTED SOURCE**: the image this repository is built in has no `rustc` / `cargo`, so this file has never been
//! compiled.  It binds exactly the symbols that the two tested host layers bind -- `bliss-rs_amd/csrc/bliss_audio.hpp`
//! (C++17, `tests/cpp/test_bliss_audio.cpp`) and `bliss-rs_amd/_ffi.py` (ctypes; `__graft_entry__.build()` checks its
//! signature table against the header and the built library) -- and the declarations below are a line-by-line
//! transcription of `include/blissgpu.h`.
//!
//! What it replaces in the crate (file:line of bliss-rs at the commit under /root/reference):
//!   * `Song::analyze_with_options`            src/song/mod.rs:403-508     -> [`analyze_with_options`]
//!   * the compute half of `Decoder::analyze_paths_with_options`
//!                                             src/song/decoder.rs:278-332 -> [`analyze_decoded`] (one device batch), or the
//!     crate's own worker threads calling [`analyze_with_options`]: concurrent calls are coalesced inside the library and
//!     spread over every GPU of the node
//!   * `FFmpegDecoder::resample_frame` (libswresample to mono 22 050 Hz f32)
//!                                             src/song/decoder/ffmpeg.rs:36-109 -> [`analyze_native`] / [`DecodedPcm`]: the
//!     decoder hands over the codec's own frames, the conversion runs on the device, bit for bit
//!   * `PreAnalyzedSong::to_song_with_options` src/song/decoder.rs:85-101  -> unchanged, it calls `Song::analyze_with_options`
//!   * `euclidean_distance` / `cosine_distance` / `mahalanobis_distance`
//!                                             src/playlist.rs:65-79,140-142 -> [`distance`], [`pairwise`]
//!   * `closest_to_songs` / `song_to_song`     src/playlist.rs:256-326     -> [`order_on_device`]
//!
//! build.rs of the crate, under the feature:
//! ```ignore
//! if std::env::var("CARGO_FEATURE_GPU").is_ok() {
//!     let dir = std::env::var("BLISSGPU_LIB_DIR").expect("BLISSGPU_LIB_DIR = directory of libblissgpu.so");
//!     println!("cargo:rustc-link-search=native={dir}");
//!     println!("cargo:rustc-link-lib=dylib=blissgpu");
//! }
//! ```
#![cfg(feature = "gpu")]
#![allow(non_camel_case_types)]

use crate::{Analysis, AnalysisOptions, BlissError, BlissResult, FeaturesVersion, Song};
use ndarray::Array2;
use std::ffi::CStr;
use std::os::raw::{c_char, c_int, c_void};

/// Raw declarations (`include/blissgpu.h`).
pub mod sys {
    use std::os::raw::{c_char, c_int, c_void};

    #[repr(C)]
    pub struct blissgpu_ctx {
        _private: [u8; 0],
    }
    #[repr(C)]
    pub struct blissgpu_node {
        _private: [u8; 0],
    }

    pub const BLISSGPU_OK: c_int = 0;
    pub const BLISSGPU_ERR_NO_DEVICE: c_int = 1;
    pub const BLISSGPU_ERR_INVALID: c_int = 2;
    pub const BLISSGPU_ERR_HIP: c_int = 3;
    pub const BLISSGPU_ERR_OOM: c_int = 4;
    pub const BLISSGPU_ERR_NAN: c_int = 5;
    pub const BLISSGPU_ERR_RCCL: c_int = 6;
    pub const BLISSGPU_ERR_TIMEOUT: c_int = 7;
    pub const BLISSGPU_SONG_OK: i32 = 0;
    pub const BLISSGPU_SONG_TOO_SHORT: i32 = 1;
    pub const BLISSGPU_METRIC_EUCLIDEAN: c_int = 0;
    pub const BLISSGPU_METRIC_COSINE: c_int = 1;
    pub const BLISSGPU_METRIC_MAHALANOBIS: c_int = 2;
    pub const BLISSGPU_SAMPLE_F32: c_int = 0;
    pub const BLISSGPU_SAMPLE_S16: c_int = 1;
    pub const BLISSGPU_SAMPLE_S32: c_int = 2;
    pub const BLISSGPU_SAMPLE_RATE: u32 = 22050;
    /// the pinned staging ring of the host PCM feed (a `Vec<f32>` is pageable memory): worker threads, 0 = off
    pub const BLISSGPU_OPT_STAGE_LANES: c_int = 9;
    pub const BLISSGPU_OPT_STAGE_SLAB_KIB: c_int = 10;
    pub const BLISSGPU_OPT_STAGE_SLABS: c_int = 11;
    pub const BLISSGPU_OPT_STAGE_NUMA: c_int = 12;

    /// One song as the decoder delivers it (host memory): `frames` frames of `channels` interleaved samples.
    #[repr(C)]
    #[derive(Clone, Copy)]
    pub struct blissgpu_decoded_song {
        pub pcm: *const c_void,
        pub frames: u64,
        pub sample_rate: u32,
        pub channels: u16,
        pub sample_format: u16,
    }

    extern "C" {
        // ---- contexts ----
        pub fn blissgpu_ctx_create(device: c_int, ctx: *mut *mut blissgpu_ctx) -> c_int;
        pub fn blissgpu_ctx_destroy(ctx: *mut blissgpu_ctx) -> c_int;
        pub fn blissgpu_ctx_synchronize(ctx: *mut blissgpu_ctx) -> c_int;
        pub fn blissgpu_ctx_set_workspace_limit(ctx: *mut blissgpu_ctx, bytes: u64) -> c_int;
        pub fn blissgpu_default_device_count() -> c_int;
        pub fn blissgpu_default_device(k: c_int) -> c_int;
        pub fn blissgpu_default_device_batches(k: c_int) -> u64;
        pub fn blissgpu_default_ctx(k: c_int, ctx: *mut *mut blissgpu_ctx) -> c_int;
        pub fn blissgpu_ctx_set_option(ctx: *mut blissgpu_ctx, option: c_int, value: i64) -> c_int;
        pub fn blissgpu_ctx_staged_bytes(ctx: *mut blissgpu_ctx) -> u64;
        pub fn blissgpu_set_single_song_timeout_ms(ms: i64) -> c_int;
        pub fn blissgpu_feature_count(features_version: u32) -> u32;
        pub fn blissgpu_feature_weights(features_version: u32, out: *mut f32) -> c_int;

        // ---- Song::analyze ----
        pub fn blissgpu_analyze(pcm: *const f32, len: u64, features_version: u32, out: *mut f32, status: *mut i32) -> c_int;
        pub fn blissgpu_analyze_interleaved(pcm: *const c_void, sample_format: c_int, channels: u32, frames: u64,
                                            features_version: u32, out: *mut f32, status: *mut i32) -> c_int;
        pub fn blissgpu_analyze_batch(pcm: *const f32, offsets: *const u64, lengths: *const u64, n_songs: u32,
                                      features_version: u32, out: *mut f32, status: *mut i32) -> c_int;
        pub fn blissgpu_analyze_batch_s16(pcm: *const i16, offsets: *const u64, lengths: *const u64, n_songs: u32,
                                          features_version: u32, out: *mut f32, status: *mut i32) -> c_int;
        pub fn blissgpu_analyze_batch_interleaved(pcm: *const c_void, sample_format: c_int, channels: u32,
                                                  offsets: *const u64, lengths: *const u64, n_songs: u32,
                                                  features_version: u32, out: *mut f32, status: *mut i32) -> c_int;
        pub fn blissgpu_analyze_decoded(pcm: *const c_void, sample_format: c_int, channels: u32, frames: u64, sample_rate: u32,
                                        features_version: u32, out: *mut f32, status: *mut i32) -> c_int;
        pub fn blissgpu_analyze_batch_decoded(songs: *const blissgpu_decoded_song, n_songs: u32, features_version: u32,
                                              out: *mut f32, status: *mut i32) -> c_int;
        pub fn blissgpu_resampled_len(frames: u64, sample_rate: u32) -> u64;
        pub fn blissgpu_default_reset() -> c_int;
        pub fn blissgpu_analyze_batch_device(ctx: *mut blissgpu_ctx, d_pcm: *const f32, offsets: *const u64,
                                             lengths: *const u64, n_songs: u32, features_version: u32, d_out: *mut f32,
                                             d_status: *mut i32) -> c_int;

        // ---- distances, playlist ordering ----
        pub fn blissgpu_distance(a: *const f32, b: *const f32, d: u32, metric: c_int, m_matrix: *const f32, out: *mut f32) -> c_int;
        pub fn blissgpu_pairwise(a: *const f32, n: u64, b: *const f32, m: u64, d: u32, metric: c_int, m_matrix: *const f32,
                                 out: *mut f32) -> c_int;
        pub fn blissgpu_set_distance(seeds: *const f32, n_seeds: u32, cand: *const f32, n: u64, d: u32, metric: c_int,
                                     m_matrix: *const f32, out: *mut f32) -> c_int;
        pub fn blissgpu_closest_to_songs(seeds: *const f32, n_seeds: u32, cand: *const f32, n: u64, d: u32, metric: c_int,
                                         m_matrix: *const f32, order: *mut u32, dist: *mut f32) -> c_int;
        pub fn blissgpu_song_to_song(seeds: *const f32, n_seeds: u32, cand: *const f32, n: u64, d: u32, metric: c_int,
                                     m_matrix: *const f32, order: *mut u32) -> c_int;

        // ---- one process, every GPU of the node ----
        pub fn blissgpu_node_create(n_devices: c_int, devices: *const c_int, node: *mut *mut blissgpu_node) -> c_int;
        pub fn blissgpu_node_destroy(node: *mut blissgpu_node) -> c_int;
        pub fn blissgpu_node_analyze(node: *mut blissgpu_node, pcm: *const f32, offsets: *const u64, lengths: *const u64,
                                     n_songs: u32, features_version: u32, out: *mut f32, status: *mut i32) -> c_int;
        pub fn blissgpu_node_pairwise(node: *mut blissgpu_node, metric: c_int, m_matrix: *const f32, out: *mut f32) -> c_int;
        pub fn blissgpu_node_synchronize(node: *mut blissgpu_node) -> c_int;
        pub fn blissgpu_shard_plan(lengths: *const u64, n_songs: u32, world: u32, rank_of_song: *mut u32) -> c_int;

        // ---- page-locked decode buffers (full PCIe rate) ----
        pub fn blissgpu_host_alloc(h_ptr: *mut *mut c_void, bytes: u64) -> c_int;
        pub fn blissgpu_host_free(h_ptr: *mut c_void) -> c_int;

        pub fn blissgpu_strerror(code: c_int) -> *const c_char;
        pub fn blissgpu_last_error() -> *const c_char;
        pub fn blissgpu_version() -> *const c_char;
    }
}

fn last_error() -> String {
    let p: *const c_char = unsafe { sys::blissgpu_last_error() };
    if p.is_null() {
        return String::new();
    }
    unsafe { CStr::from_ptr(p) }.to_string_lossy().into_owned()
}

/// A failed call never crosses the boundary as a panic: whole-call return code + per-song status.
fn gpu_err(rc: c_int) -> BlissError {
    match rc {
        sys::BLISSGPU_ERR_NO_DEVICE => BlissError::ProviderError(format!("no usable MI355X for libblissgpu (there is no CPU fallback): {}", last_error())),
        _ => BlissError::AnalysisError(format!("blissgpu error {rc}: {}", last_error())),
    }
}

fn too_short() -> BlissError {
    // src/song/mod.rs:426-430
    BlissError::AnalysisError(String::from("empty or too short song."))
}

fn version_code(v: FeaturesVersion) -> u32 {
    v as u16 as u32 // FeaturesVersion::Version1 = 1, Version2 = 2 (src/lib.rs:151-160)
}

/// Same contract as the CPU `Song::analyze_with_options` (src/song/mod.rs:413-508): mono 22 050 Hz f32 samples in,
/// `Analysis` out, `AnalysisError("empty or too short song.")` below 8192 samples.  Called from the crate's
/// `number_cores` worker threads (src/song/decoder.rs:299-329) it keeps every GPU of the node busy: the library coalesces
/// concurrent calls into device batches, one default context per visible device.
pub fn analyze_with_options(sample_array: &[f32], analysis_options: &AnalysisOptions) -> BlissResult<Analysis> {
    let version = analysis_options.features_version;
    let mut out = vec![0f32; version.feature_count()];
    let mut status = 0i32;
    let rc = unsafe {
        sys::blissgpu_analyze(sample_array.as_ptr(), sample_array.len() as u64, version_code(version), out.as_mut_ptr(), &mut status)
    };
    if rc != sys::BLISSGPU_OK {
        return Err(gpu_err(rc));
    }
    if status == sys::BLISSGPU_SONG_TOO_SHORT {
        return Err(too_short());
    }
    Analysis::new(out, version)
}

impl Song {
    /// Drop-in for the CPU associated function of the same name when the `gpu` feature is on.
    pub fn analyze_with_options_gpu(sample_array: &[f32], analysis_options: &AnalysisOptions) -> BlissResult<Analysis> {
        analyze_with_options(sample_array, analysis_options)
    }
}

/// Bulk form for `Decoder::analyze_paths_with_options`: decode on the CPU workers as today, then hand the decoded buffers
/// over in ONE call instead of analysing per thread (src/song/decoder.rs:304-328).  The library orders the songs by
/// length, cuts them into chunks that fit its workspace and streams the PCM over PCIe group by group -- straight from
/// the callers' buffers: nothing is copied on the host.
pub fn analyze_decoded(songs: &[&[f32]], version: FeaturesVersion) -> BlissResult<Vec<BlissResult<Analysis>>> {
    let native: Vec<DecodedPcm> = songs.iter().map(|s| DecodedPcm::F32 { samples: s, channels: 1, sample_rate: sys::BLISSGPU_SAMPLE_RATE }).collect();
    analyze_native(&native, version)
}

/// What a decoder delivers BEFORE the conversion `FFmpegDecoder::resample_frame` does (src/song/decoder/ffmpeg.rs:36-109):
/// interleaved frames at the file's own rate.  The library converts to mono 22 050 Hz f32 on the device exactly as
/// libswresample does with the decoder's options -- same filter, same summation order; the reference's Adler-32 decoder
/// tests (ffmpeg.rs:433-452) hold for its output -- so a decoder built on this skips `resample_frame` altogether.
#[derive(Clone, Copy)]
pub enum DecodedPcm<'a> {
    F32 { samples: &'a [f32], channels: u16, sample_rate: u32 },
    S16 { samples: &'a [i16], channels: u16, sample_rate: u32 },
    /// FFmpeg's AV_SAMPLE_FMT_S32 (24-bit streams arrive left-justified)
    S32 { samples: &'a [i32], channels: u16, sample_rate: u32 },
}

impl<'a> DecodedPcm<'a> {
    fn raw(&self) -> sys::blissgpu_decoded_song {
        let (pcm, len, channels, sample_rate, fmt) = match *self {
            DecodedPcm::F32 { samples, channels, sample_rate } => (samples.as_ptr() as *const c_void, samples.len(), channels, sample_rate, sys::BLISSGPU_SAMPLE_F32),
            DecodedPcm::S16 { samples, channels, sample_rate } => (samples.as_ptr() as *const c_void, samples.len(), channels, sample_rate, sys::BLISSGPU_SAMPLE_S16),
            DecodedPcm::S32 { samples, channels, sample_rate } => (samples.as_ptr() as *const c_void, samples.len(), channels, sample_rate, sys::BLISSGPU_SAMPLE_S32),
        };
        sys::blissgpu_decoded_song { pcm, frames: (len / channels.max(1) as usize) as u64, sample_rate, channels, sample_format: fmt as u16 }
    }
    /// Samples at 22 050 Hz this song becomes (`PreAnalyzedSong::duration`, src/song/decoder.rs:34-65).
    pub fn resampled_len(&self) -> u64 {
        let r = self.raw();
        unsafe { sys::blissgpu_resampled_len(r.frames, r.sample_rate) }
    }
}

/// `analyze_paths_with_options` for decoders that keep the file's format: a library of 44.1 / 48 kHz, mono / stereo, 16 /
/// 24-bit files in one call (blissgpu_analyze_batch_decoded).
pub fn analyze_native(songs: &[DecodedPcm], version: FeaturesVersion) -> BlissResult<Vec<BlissResult<Analysis>>> {
    let d = version.feature_count();
    let raw: Vec<sys::blissgpu_decoded_song> = songs.iter().map(|s| s.raw()).collect();
    let mut out = vec![0f32; songs.len() * d];
    let mut status = vec![0i32; songs.len()];
    let rc = unsafe { sys::blissgpu_analyze_batch_decoded(raw.as_ptr(), raw.len() as u32, version_code(version), out.as_mut_ptr(), status.as_mut_ptr()) };
    if rc != sys::BLISSGPU_OK {
        return Err(gpu_err(rc));
    }
    Ok(status
        .iter()
        .enumerate()
        .map(|(i, &st)| if st == sys::BLISSGPU_SONG_TOO_SHORT { Err(too_short()) } else { Analysis::new(out[i * d..(i + 1) * d].to_vec(), version) })
        .collect())
}

/// The three metrics the library evaluates (bit for bit like src/playlist.rs:65-79,140-142, ndarray's summation order
/// included).  Custom closures and the isolation-forest metric stay on the CPU iterators.
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum Metric {
    Euclidean,
    Cosine,
    Mahalanobis,
}

impl Metric {
    fn code(self) -> c_int {
        match self {
            Metric::Euclidean => sys::BLISSGPU_METRIC_EUCLIDEAN,
            Metric::Cosine => sys::BLISSGPU_METRIC_COSINE,
            Metric::Mahalanobis => sys::BLISSGPU_METRIC_MAHALANOBIS,
        }
    }
}

fn matrix_ptr(m: Option<&Array2<f32>>) -> (Option<Array2<f32>>, *const f32) {
    let owned = m.map(|m| m.as_standard_layout().to_owned());
    let p = owned.as_ref().map_or(std::ptr::null(), |m| m.as_ptr());
    (owned, p)
}

/// One pair (`Analysis::distance` / `Song::distance`, src/song/mod.rs:364-370,519-521): one kernel launch.
pub fn distance(a: &[f32], b: &[f32], metric: Metric, m: Option<&Array2<f32>>) -> BlissResult<f32> {
    assert_eq!(a.len(), b.len());
    let (_keep, mp) = matrix_ptr(m);
    let mut out = 0f32;
    let rc = unsafe { sys::blissgpu_distance(a.as_ptr(), b.as_ptr(), a.len() as u32, metric.code(), mp, &mut out) };
    if rc != sys::BLISSGPU_OK {
        return Err(gpu_err(rc));
    }
    Ok(out)
}

/// All pairs between the rows of `a` and the rows of `b` (row-major `n x d` / `m x d`), `n x m` out.
pub fn pairwise(a: &Array2<f32>, b: &Array2<f32>, metric: Metric, m: Option<&Array2<f32>>) -> BlissResult<Array2<f32>> {
    assert_eq!(a.ncols(), b.ncols());
    let (a, b) = (a.as_standard_layout(), b.as_standard_layout());
    let (_keep, mp) = matrix_ptr(m);
    let mut out = vec![0f32; a.nrows() * b.nrows()];
    let rc = unsafe {
        sys::blissgpu_pairwise(a.as_ptr(), a.nrows() as u64, b.as_ptr(), b.nrows() as u64, a.ncols() as u32, metric.code(), mp, out.as_mut_ptr())
    };
    if rc != sys::BLISSGPU_OK {
        return Err(gpu_err(rc));
    }
    Ok(Array2::from_shape_vec((a.nrows(), b.nrows()), out).expect("n x m"))
}

/// `closest_to_songs` (src/playlist.rs:256-270; `chain = false`: stable sort by the summed distance to the seed set) and
/// `song_to_song` (:272-326; `chain = true`: greedy nearest-neighbour chain) for the built-in metrics: the same order as
/// the CPU iterators, ties included; only the permutation comes back from the device.  A NaN distance is
/// `BLISSGPU_ERR_NAN` here where `n32()` / `argmin().unwrap()` panic on the CPU path.
pub fn order_on_device<T: AsRef<Song> + Clone>(initial: &[T], candidates: &[T], metric: Metric, m: Option<&Array2<f32>>,
                                               chain: bool) -> BlissResult<Vec<T>> {
    let d = candidates.first().map_or(0, |s| s.as_ref().analysis.as_vec().len());
    let flat = |songs: &[T]| songs.iter().flat_map(|s| s.as_ref().analysis.as_vec()).collect::<Vec<f32>>();
    let (seeds, cand) = (flat(initial), flat(candidates));
    let (_keep, mp) = matrix_ptr(m);
    let mut order = vec![0u32; candidates.len()];
    let rc = unsafe {
        if chain {
            sys::blissgpu_song_to_song(seeds.as_ptr(), initial.len() as u32, cand.as_ptr(), candidates.len() as u64, d as u32,
                                       metric.code(), mp, order.as_mut_ptr())
        } else {
            sys::blissgpu_closest_to_songs(seeds.as_ptr(), initial.len() as u32, cand.as_ptr(), candidates.len() as u64, d as u32,
                                           metric.code(), mp, order.as_mut_ptr(), std::ptr::null_mut())
        }
    };
    if rc != sys::BLISSGPU_OK {
        return Err(gpu_err(rc));
    }
    Ok(order.into_iter().map(|i| candidates[i as usize].clone()).collect())
}

/// One process driving every GPU of the node: songs sharded by sample count, one `ncclAllGather` of the feature rows over
/// xGMI, row-block-sharded pairwise distances.  (The worker-thread pattern above needs none of this.)
pub struct Node {
    raw: *mut sys::blissgpu_node,
    n_songs: usize,
}

// the library serialises per context and is re-entrant; the handle may move between threads
unsafe impl Send for Node {}

impl Node {
    /// `devices = None`: HIP devices 0 .. n_devices.
    pub fn new(n_devices: usize, devices: Option<&[c_int]>) -> BlissResult<Node> {
        let mut raw = std::ptr::null_mut();
        let rc = unsafe { sys::blissgpu_node_create(n_devices as c_int, devices.map_or(std::ptr::null(), |d| d.as_ptr()), &mut raw) };
        if rc != sys::BLISSGPU_OK {
            return Err(gpu_err(rc));
        }
        Ok(Node { raw, n_songs: 0 })
    }

    /// Analyses `songs` across the node; afterwards every device holds the full `n x d` matrix.
    pub fn analyze(&mut self, songs: &[&[f32]], version: FeaturesVersion) -> BlissResult<Vec<BlissResult<Analysis>>> {
        let d = version.feature_count();
        let lengths: Vec<u64> = songs.iter().map(|s| s.len() as u64).collect();
        let mut offsets = Vec::with_capacity(songs.len());
        let mut pcm: Vec<f32> = Vec::new();
        for s in songs {
            offsets.push(pcm.len() as u64);
            pcm.extend_from_slice(s);
        }
        let mut out = vec![0f32; songs.len() * d];
        let mut status = vec![0i32; songs.len()];
        let rc = unsafe {
            sys::blissgpu_node_analyze(self.raw, pcm.as_ptr(), offsets.as_ptr(), lengths.as_ptr(), songs.len() as u32, version_code(version),
                                       out.as_mut_ptr(), status.as_mut_ptr())
        };
        if rc != sys::BLISSGPU_OK {
            return Err(gpu_err(rc));
        }
        self.n_songs = songs.len();
        Ok(status
            .iter()
            .enumerate()
            .map(|(i, &st)| if st == sys::BLISSGPU_SONG_TOO_SHORT { Err(too_short()) } else { Analysis::new(out[i * d..(i + 1) * d].to_vec(), version) })
            .collect())
    }

    /// All-pairs distances over the gathered matrix, one row block per device.
    pub fn pairwise(&mut self, metric: Metric, m: Option<&Array2<f32>>) -> BlissResult<Array2<f32>> {
        let (_keep, mp) = matrix_ptr(m);
        let n = self.n_songs;
        let mut out = vec![0f32; n * n];
        let rc = unsafe { sys::blissgpu_node_pairwise(self.raw, metric.code(), mp, out.as_mut_ptr()) };
        if rc != sys::BLISSGPU_OK {
            return Err(gpu_err(rc));
        }
        Ok(Array2::from_shape_vec((n, n), out).expect("n x n"))
    }
}

impl Drop for Node {
    fn drop(&mut self) {
        unsafe { sys::blissgpu_node_destroy(self.raw) };
    }
}
