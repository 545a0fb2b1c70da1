// internal.hpp -- shared declarations of the MI355X analysis library (not part of the C ABI).
//
// Vocabulary follows the reference (bliss-rs): songs, frames, descriptors, features.
//   timbral frames : PVoc windows, W=512 hop=128   (src/timbral.rs:40-41, src/aubio.rs:182-264)
//   tempo frames   : PVocTempo windows, W=512 hop=256 (src/temporal.rs:40-41, src/aubio.rs:338-425)
//   chroma frames  : STFT windows, W=8192 hop=2205 (src/chroma.rs:39,74, src/utils.rs:26-64)
//   loudness chunks: 1024 samples                  (src/misc.rs:44)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bg {

constexpr uint32_t SAMPLE_RATE = 22050;   // src/lib.rs:140
constexpr int W512 = 512;                 // SpectralDesc::WINDOW_SIZE / BPMDesc::WINDOW_SIZE
constexpr int HOP_T = 128;                // SpectralDesc::HOP_SIZE
constexpr int HOP_B = 256;                // BPMDesc::HOP_SIZE
constexpr int W8192 = 8192;               // ChromaDesc::WINDOW_SIZE
constexpr int HOP_C = 2205;               // src/chroma.rs:74
constexpr int CBINS = W8192 / 2 + 1;      // 4097
constexpr int CBINS_PAD = 4128;           // row pitch of the stored spectrogram: 129 lines of 128 bytes, so every row starts on a line
constexpr int BANK_PITCH = CBINS_PAD;     // row pitch of the filter bank (zero beyond bin 4096)
constexpr int BANK_ROWS = 12;             // chroma classes
constexpr int LOUD_W = 1024;              // LoudnessDesc::WINDOW_SIZE
constexpr int MIN_SAMPLES = 8192;         // src/song/mod.rs:417-430
constexpr int N_TUNING = 100;             // pitch_tuning histogram bins at resolution 0.01
constexpr int PIP_LO = 57, PIP_HI = 1483; // centre bins visited by pip_track at n_fft=8192 (chroma.rs:302-313)
constexpr int PIP_MAX_PER_FRAME = 714;    // peaks cannot be adjacent: ceil(1427/2)
constexpr int CAND_BUDGET_PER_FRAME = 48; // tuning-candidate pool of a chunk: slots per chroma frame (white noise needs ~8; see tune_select_kernel)
// Coarse magnitude histogram of the tuning estimate: bin = f32 bit pattern >> COARSE_SHIFT, i.e. 2^(23 - COARSE_SHIFT) bins
// per octave.  64 per octave: the peaks that share the median's bins -- the candidates tune_pass2_kernel must evaluate in
// f64 from the spectrogram -- are half as many as with 32 (2.2 % of all peaks), and 14 bits are what a 32-bit peak record
// has left beside the centre bin and the pitch bin.
#ifndef COARSE_SHIFT
#define COARSE_SHIFT 17
#endif
constexpr uint32_t COARSE_LOW_MASK = (1u << COARSE_SHIFT) - 1u;
constexpr int H1_BINS = 1 << (31 - COARSE_SHIFT);   // every positive f32 has a bin
constexpr int BT_WINLEN = 512, BT_STEP = 128, BT_LAGLEN = 128;  // src/aubio.rs:1337-1341, 920-922
constexpr int BT_PRE_STRIDE = 264;  // floats per beat-tracker run written by beat_acf_kernel (kernels_tempo.hip)
constexpr int F512_TILE = 512;            // FFT-512 frames per workgroup (16 lane-groups x 32 consecutive frames + 1 halo frame each)
constexpr int CH_TILE = 64;               // chroma frames per workgroup in the contraction kernel
constexpr int STFT_TILE = 16;             // chroma frames per workgroup in the STFT kernel

// Per-song descriptor, built on the host, read by every kernel.
struct SongDesc {
    uint64_t pcm_off;   // first sample of the song in the batch PCM buffer
    uint64_t n;         // samples
    uint64_t t_off;     // first timbral frame in the batch-wide series arrays
    uint64_t b_off;     // first tempo frame
    uint64_t c_off;     // first chroma frame
    uint64_t e_off;     // first 256-sample energy block
    uint32_t n_t;       // timbral frames   floor((n-512)/128)+1
    uint32_t n_b;       // tempo frames     floor((n-512)/256)+1
    uint32_t n_f;       // FFT-512 frames actually computed = max(n_t, 2*n_b)
    uint32_t n_c;       // chroma frames    min(ceil_f32(n/2205), n/2205+1)
    uint32_t n_e;       // 256-sample blocks ceil(n/256)
    uint32_t n_l;       // loudness chunks  ceil(n/1024)
    uint32_t row;       // output row
    uint32_t ok;        // 1 = analyse, 0 = too short (status already set by the host)
};

// Frames whose rolloff bin the FFT-512 kernel could not prove (see its epilogue) are handed to rolloff_fix_kernel: the
// frame's 256 magnitudes at the entry of its own index, and ROLLOFF_UNPROVEN (no rolloff is negative) in its place of the
// rolloff series, where the exact pass finds it.  The record lives in device memory (filled by the host per chunk); the
// entries borrow the memory of the spectrogram and the peak records, dead until the FFT-8192 kernel starts -- room for every
// frame of the chunk.
struct RollFix {
    float* mags;        // [cap][256]: entry t = the magnitudes of the chunk's timbral frame t (written for unproven frames only)
    uint32_t cap;       // = the chunk's timbral frames
};
constexpr float ROLLOFF_UNPROVEN = -1.0f;

// Per-song scalars produced by the tuning stage.
struct TuningState {
    uint32_t n_peaks;      // total pip_track peaks
    uint32_t b_lo, b_hi;   // coarse bins holding the two middle order statistics
    uint32_t below;        // peaks in coarse bins < b_lo
    uint32_t n_cand;       // candidates appended so far
    int32_t tuning_idx;    // argmax bin (first max); tuning = -0.5 + 0.01*idx ; -1 => tuning 0.0 (no peaks)
    uint32_t cand_off;     // first slot of the song's candidates in the chunk's pool
    uint32_t cand_cap;     // slots granted = peaks inside [b_lo, b_hi]; 0 => the pool was exhausted: tune_final re-scans the records
};

// Beat-tracker result per song
struct TempoState {
    float tempo;        // normalised feature
    uint32_t n_bpms;
};

struct DeviceTables {   // constant tables, built once per context
    const float2* tw8192;     // exp(-2*pi*i*k/8192), k < 8192
    const float2* tw512;      // exp(-2*pi*i*k/512),  k < 512
    const float* hann8192;    // periodic Hann, src/utils.rs:37-39
    const float* hannz512;    // hanningz, src/aubio.rs:151-154
    const double* chroma_bank;// [N_TUNING+1][BANK_ROWS][BANK_PITCH] chroma filters (zero padded); slot N_TUNING = tuning 0.0
    const float* bt_rwv;      // [128] Rayleigh weighting, src/aubio.rs:925-930
    const float* bt_dfwv;     // [512] detection-function weighting, src/aubio.rs:933-936
};

enum KernelId : int {
    K_FFT512 = 0,
    K_ONSET,
    K_BEAT,
    K_STFT8192,
    K_TUNE_SELECT,
    K_TUNE_PASS2,
    K_TUNE_FINAL,
    K_CHROMA,
    K_SUMMARY,
    K_FINALIZE,
    K_PAIRWISE, K_SET_DISTANCE, K_SONG_TO_SONG,
    K_SYNTH,
    K_ROLLFIX,
    K_COUNT
};

// lower_bound on a prefix array: largest s with prefix[s] <= x  (prefix has n+1 entries, prefix[0]=0)
__device__ __forceinline__ uint32_t find_segment(const uint32_t* __restrict__ prefix, uint32_t n, uint32_t x) {
    uint32_t lo = 0, hi = n;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (prefix[mid] <= x) lo = mid; else hi = mid;
    }
    return lo;
}

// ---- launchers (one per .hip file) ----
struct Workspace {
    // timbral series [total_t]
    float *centroid, *rolloff, *flatness;
    // tempo [total_b]
    float *flux, *thresholded;
    // 256-sample energy blocks [total_e], zero-crossings per block
    float* e256;
    uint32_t* zc256;
    // chroma
    float* spec;            // [total_c][CBINS_PAD] magnitudes (f32, exactly what the reference widens to f64)
    float* frame_max;       // [total_c]
    uint32_t* h1;           // [n_songs][H1_BINS]
    uint32_t* hist100;      // [n_songs][N_TUNING]
    TuningState* tuning;    // [n_songs]
    uint32_t* peak_rec;     // [total_c][PIP_MAX_PER_FRAME] per-peak records written by the STFT kernel (see peak_record())
    uint32_t* peak_cnt;     // [total_c] records per frame
    double* cand_mag;       // [cand_cap] candidate pool of the chunk (slots handed out by tune_select_kernel)
    uint8_t* cand_pb;       // [cand_cap]
    uint32_t* cand_cursor;  // [0] next free pool slot
    uint32_t cand_cap;
    double* chroma_part;    // [total chroma tiles][10] partial sums of interval features
    double* dbg_chroma;     // [total_c][12] chroma_stft's normalised columns, or NULL (BLISSGPU_OPT_DEBUG_CHROMA)
    double* dbg_interval;   // [n_songs][10] interval means before the normalisation, or NULL
    TempoState* tempo;      // [n_songs]
    float* run_bpm;         // [n_songs][runs_pitch] bpm after each beat-tracker run
    uint32_t* run_cnt;      // [n_songs][runs_pitch] beats recorded while that bpm was current
    uint32_t runs_pitch;
    float* bt_pre;          // [total_b / 128 + n_songs][BT_PRE_STRIDE] per-run records; song s starts at run b_off / 128 + s
    float* summary;         // [n_songs][16] features 1..9 (zcr, timbral, loudness summaries)
    RollFix* roll_fix;      // the exact rolloff pass's record (see RollFix)
    size_t roll_fix_bytes;  // the borrowed stretch: spectrogram + peak records
};

struct Batch {
    const float* pcm;
    const SongDesc* songs;    // device
    uint32_t n_songs;
    // tile prefix arrays (device), n_songs+1 entries each
    const uint32_t* pfx_f;    // fft512 tiles
    const uint32_t* pfx_c;    // STFT tiles (STFT_TILE chroma frames each)
    const uint32_t* pfx_ct;   // 64-frame chroma tiles (tuning pass 2 workgroups, chroma_part slots)
    const uint32_t* pfx_cw;   // chroma contraction workgroups (4 tiles of 64 frames each)
    uint32_t tiles_f, tiles_c, tiles_ct, tiles_cw;
    uint64_t total_b;         // tempo frames in the batch
    uint32_t max_nb;          // longest song's tempo-frame count
    uint32_t max_nt;          // longest song's timbral-frame count
};

void launch_fft512(const Batch&, const Workspace&, const DeviceTables&, hipStream_t, bool rolloff_exact_all = false, bool seq_flux = false);
void launch_rolloff_fix(const Batch&, const Workspace&, uint64_t total_t, hipStream_t);
void launch_onset(const Batch&, const Workspace&, hipStream_t);
void launch_beat(const Batch&, const Workspace&, const DeviceTables&, hipStream_t);
// the two halves of launch_beat on streams of their own (the masked tail): autocorrelations (parallel), state machines (a wavefront per song)
void launch_beat_acf(const Batch&, const Workspace&, const DeviceTables&, hipStream_t);
void launch_beat_track(const Batch&, const Workspace&, const DeviceTables&, hipStream_t);
void launch_stft8192(const Batch&, const Workspace&, const DeviceTables&, hipStream_t, int shape = 0);  // shape: BLISSGPU_OPT_STFT_SHAPE
// a contiguous range of a chunk's songs [s0, s1) with the tile / workgroup ranges that belong to it in pfx_ct / pfx_cw
// units (the tuning estimate and the contraction of a one-chunk batch run in two halves; NULL = the whole chunk)
struct SongRange { uint32_t s0, s1, ct0, ct1, cw0, cw1; };
void launch_tune_select(const Batch&, const Workspace&, hipStream_t, const SongRange* r = nullptr);
void launch_tune_pass2(const Batch&, const Workspace&, hipStream_t, const SongRange* r = nullptr);
void launch_tune_final(const Batch&, const Workspace&, hipStream_t, const SongRange* r = nullptr);
void launch_chroma(const Batch&, const Workspace&, const DeviceTables&, hipStream_t, const SongRange* r = nullptr);
void launch_summary(const Batch&, const Workspace&, hipStream_t);
void launch_finalize(const Batch&, const Workspace&, uint32_t features_version, float* d_out, int32_t* d_status,
                     int32_t* dbg_tuning, uint32_t* dbg_nbpms, hipStream_t);
void launch_chroma_bank(double* bank, hipStream_t);
// raw decoder output at 22 050 Hz -> mono f32: sample_format BLISSGPU_SAMPLE_* (s16: sample / 32768, s32: sample / 2^31),
// `channels` interleaved
void launch_pcm_convert(const void* in, int sample_format, uint32_t channels, float* out, uint64_t frames, hipStream_t st);
// raw decoder output at another rate -> mono 22 050 Hz f32 (libswresample's default resampler, resample.hpp)
struct SwrPlan;
hipError_t launch_resample(const void* in, int sample_format, uint32_t channels, uint64_t frames, const SwrPlan& plan,
                           const float* d_bank, float* out, uint64_t n_out, hipStream_t st);
// one pair distance with both vectors passed by value (no staging copies); result -> *out (device-visible host word)
void launch_pair_distance(const float* a, const float* b, uint32_t d, int metric, const float* d_M, float* out, hipStream_t st);
// playlist ordering (kernels_playlist.hip)
void launch_set_distance(const float* seeds, uint32_t n_seeds, const float* cand, uint64_t n, uint32_t d, int metric,
                         const float* M, float* dist, uint32_t* keys, uint32_t* idx, uint32_t* nan_flag, hipStream_t st);
// stable radix sort of (key, value) pairs; tmp == NULL reports the scratch bytes; keys_in / vals_in are scratch afterwards
// (sync: one zeroed word -> the single-launch form when the tiles fit max_coresident workgroups; NULL: one launch per step)
hipError_t sort_pairs_u32(void* tmp, size_t* tmp_bytes, uint32_t* keys_in, uint32_t* keys_out, uint32_t* vals_in,
                          uint32_t* vals_out, uint32_t n, hipStream_t st, uint32_t* sync = nullptr, uint32_t max_coresident = 0);
void launch_song_to_song(const float* seeds, uint32_t n_seeds, const float* cand, uint32_t n, uint32_t d, int metric,
                         const float* M, uint32_t* order, unsigned long long* slots, uint32_t* sync, uint32_t grid,
                         hipStream_t st);
void launch_pairwise(const float* A, uint64_t n, const float* B, uint64_t m, uint32_t d, int metric, const float* M,
                     int m_is_diag, float* out, uint64_t ld_out, hipStream_t);
void launch_synth(float* pcm, const SongDesc* songs, uint32_t n_songs, const uint32_t* pfx_e, uint32_t tiles_e,
                  uint32_t first_song_index, const uint32_t* d_song_index, hipStream_t);

}  // namespace bg
