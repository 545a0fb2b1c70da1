// bliss_audio.hpp -- C++17 host-side mirror of the bliss-rs interface for the analysis hot path,
// implemented on top of the C ABI in include/blissgpu.h (header-only; link with -lblissgpu).
//
// The reference is a Rust crate and this image has no Rust toolchain, so this header is the compiled
// host layer that is tested here; the Rust binding a bliss-rs maintainer would add is in
// INTEGRATION.md.  Names, argument meaning and error behaviour follow the reference:
//
//   bliss::Song::analyze / analyze_with_options      src/song/mod.rs:403-508
//   bliss::Analysis, AnalysisIndex, FeaturesVersion  src/song/mod.rs:102-371, src/lib.rs:151-187
//   bliss::BlissError {Decoding,Analysis,Provider}   src/lib.rs:236-252
//   bliss::Decoder (decode / song_from_path / analyze_paths)   src/song/decoder.rs:115-333
//   bliss::euclidean_distance / cosine_distance / mahalanobis_distance   src/playlist.rs:65-142
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <functional>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <variant>
#include <vector>

#include "../../include/blissgpu.h"

namespace bliss {

constexpr uint32_t SAMPLE_RATE = 22050;  // src/lib.rs:140
constexpr uint16_t CHANNELS = 1;         // src/lib.rs:137

// ---- BlissError (src/lib.rs:236-252): value-returned on the analysis path, thrown here ----
struct BlissError : std::runtime_error {
    enum class Kind { DecodingError, AnalysisError, ProviderError } kind;
    std::string message;
    BlissError(Kind k, std::string m) : std::runtime_error(prefix(k) + m), kind(k), message(std::move(m)) {}
    static std::string prefix(Kind k) {
        switch (k) {
            case Kind::DecodingError: return "error happened while decoding file - ";
            case Kind::AnalysisError: return "error happened while analyzing file - ";
            default: return "error happened with the music library provider - ";
        }
    }
    bool operator==(const BlissError& o) const { return kind == o.kind && message == o.message; }
};
inline BlissError AnalysisError(std::string m) { return BlissError(BlissError::Kind::AnalysisError, std::move(m)); }
inline BlissError DecodingError(std::string m) { return BlissError(BlissError::Kind::DecodingError, std::move(m)); }
inline BlissError ProviderError(std::string m) { return BlissError(BlissError::Kind::ProviderError, std::move(m)); }

// a failure of the GPU library itself (no device, HIP error): not a BlissError, there is no CPU path
struct GpuError : std::runtime_error {
    int code;
    GpuError(int c) : std::runtime_error(std::string("blissgpu: ") + blissgpu_strerror(c) + " (" + blissgpu_last_error() + ")"), code(c) {}
};
inline void check(int rc) { if (rc != BLISSGPU_OK) throw GpuError(rc); }

// ---- FeaturesVersion (src/lib.rs:151-187) ----
enum class FeaturesVersion : uint16_t { Version1 = 1, Version2 = 2 };
constexpr FeaturesVersion LATEST = FeaturesVersion::Version2;
constexpr size_t feature_count(FeaturesVersion v) { return v == FeaturesVersion::Version2 ? 23 : 20; }
constexpr size_t NUMBER_FEATURES = feature_count(LATEST);  // src/song/mod.rs:222
inline FeaturesVersion features_version_try_from(uint16_t v) {
    if (v == 1 || v == 2) return static_cast<FeaturesVersion>(v);
    throw ProviderError("This features' version (" + std::to_string(v) + ") does not exist");
}
inline std::vector<float> feature_weights(FeaturesVersion v) {  // row-major d x d
    std::vector<float> m(feature_count(v) * feature_count(v));
    check(blissgpu_feature_weights(static_cast<uint32_t>(v), m.data()));
    return m;
}

// ---- AnalysisIndex (src/song/mod.rs:102-156) ----
enum class AnalysisIndex : size_t {
    Tempo, Zcr, MeanSpectralCentroid, StdDeviationSpectralCentroid, MeanSpectralRolloff, StdDeviationSpectralRolloff,
    MeanSpectralFlatness, StdDeviationSpectralFlatness, MeanLoudness, StdDeviationLoudness,
    Chroma1, Chroma2, Chroma3, Chroma4, Chroma5, Chroma6, Chroma7, Chroma8, Chroma9, Chroma10, Chroma11, Chroma12, Chroma13
};

struct AnalysisOptions {  // src/song/mod.rs:248-269
    FeaturesVersion features_version = LATEST;
    size_t number_cores = 1;  // kept for interface parity; the batch is scheduled on the GPU
};

// ---- distances (src/playlist.rs:65-79, 129-142) ----
inline float distance(const std::vector<float>& a, const std::vector<float>& b, int metric, const float* m = nullptr) {
    if (a.size() != b.size()) throw std::invalid_argument("vectors must have the same length");
    float out = 0.0f;
    check(blissgpu_distance(a.data(), b.data(), static_cast<uint32_t>(a.size()), metric, m, &out));
    return out;
}
inline float euclidean_distance(const std::vector<float>& a, const std::vector<float>& b) { return distance(a, b, BLISSGPU_METRIC_EUCLIDEAN); }
inline float cosine_distance(const std::vector<float>& a, const std::vector<float>& b) { return distance(a, b, BLISSGPU_METRIC_COSINE); }
inline float mahalanobis_distance(const std::vector<float>& a, const std::vector<float>& b, const std::vector<float>& m) {
    return distance(a, b, BLISSGPU_METRIC_MAHALANOBIS, m.data());
}
using DistanceFn = std::function<float(const std::vector<float>&, const std::vector<float>&)>;
inline DistanceFn mahalanobis_distance_builder(std::vector<float> m) {
    return [m = std::move(m)](const std::vector<float>& a, const std::vector<float>& b) { return mahalanobis_distance(a, b, m); };
}
inline DistanceFn distance_metric(FeaturesVersion v) { return mahalanobis_distance_builder(feature_weights(v)); }  // src/lib.rs:176-178

// ---- Analysis (src/song/mod.rs:238-371) ----
class Analysis {
  public:
    std::vector<float> internal_analysis;
    FeaturesVersion features_version = LATEST;
    Analysis() = default;
    Analysis(std::vector<float> analysis, FeaturesVersion v) : internal_analysis(std::move(analysis)), features_version(v) {
        if (internal_analysis.size() != feature_count(v))
            throw ProviderError("Feature count " + std::to_string(internal_analysis.size()) +
                                " does not match the expected version feature count " + std::to_string(feature_count(v)));
    }
    const std::vector<float>& as_vec() const { return internal_analysis; }
    float operator[](AnalysisIndex i) const {
        if (features_version != LATEST) throw std::logic_error("Tried to index features with incompatible indexes");
        return internal_analysis[static_cast<size_t>(i)];
    }
    bool operator==(const Analysis& o) const { return features_version == o.features_version && internal_analysis == o.internal_analysis; }
    // default distance for the FeaturesVersion; the reference panics on mismatched versions (:364-370)
    float distance(const Analysis& other) const {
        if (features_version != other.features_version) throw std::logic_error("Mismatched features version between two songs or analysis");
        return distance_metric(features_version)(internal_analysis, other.internal_analysis);
    }
};

template <typename T>
using BlissResult = std::variant<T, BlissError>;

// Bulk Song::analyze_with_options over songs already decoded to mono 22 050 Hz f32 (one GPU batch).
inline std::vector<BlissResult<Analysis>> analyze_batch(const std::vector<std::vector<float>>& songs, const AnalysisOptions& opt = {}) {
    const uint32_t n = static_cast<uint32_t>(songs.size());
    const size_t d = feature_count(opt.features_version);
    std::vector<uint64_t> off(n), len(n);
    uint64_t total = 0;
    for (uint32_t i = 0; i < n; i++) { off[i] = total; len[i] = songs[i].size(); total += len[i]; }
    std::vector<float> pcm(total ? total : 1);
    for (uint32_t i = 0; i < n; i++) std::copy(songs[i].begin(), songs[i].end(), pcm.begin() + off[i]);
    std::vector<float> out(n * d);
    std::vector<int32_t> status(n);
    if (n) check(blissgpu_analyze_batch(pcm.data(), off.data(), len.data(), n, static_cast<uint32_t>(opt.features_version), out.data(), status.data()));
    std::vector<BlissResult<Analysis>> res;
    res.reserve(n);
    for (uint32_t i = 0; i < n; i++) {
        if (status[i] == BLISSGPU_SONG_OK) res.emplace_back(Analysis(std::vector<float>(out.begin() + i * d, out.begin() + (i + 1) * d), opt.features_version));
        else if (status[i] == BLISSGPU_SONG_TOO_SHORT) res.emplace_back(AnalysisError("empty or too short song."));
        else res.emplace_back(AnalysisError("analysis failed with status " + std::to_string(status[i])));
    }
    return res;
}

// Same for decoders that deliver s16 mono 22 050 Hz PCM: 2 bytes per sample over PCIe, widened on the device exactly
// like FFmpeg's s16 -> flt conversion (sample / 32768, src/song/decoder/ffmpeg.rs:36-109).
inline std::vector<BlissResult<Analysis>> analyze_batch(const std::vector<std::vector<int16_t>>& songs, const AnalysisOptions& opt = {}) {
    const uint32_t n = static_cast<uint32_t>(songs.size());
    const size_t d = feature_count(opt.features_version);
    std::vector<uint64_t> off(n), len(n);
    uint64_t total = 0;
    for (uint32_t i = 0; i < n; i++) { off[i] = total; len[i] = songs[i].size(); total += len[i]; }
    std::vector<int16_t> pcm(total ? total : 1);
    for (uint32_t i = 0; i < n; i++) std::copy(songs[i].begin(), songs[i].end(), pcm.begin() + off[i]);
    std::vector<float> out(n * d);
    std::vector<int32_t> status(n);
    if (n) check(blissgpu_analyze_batch_s16(pcm.data(), off.data(), len.data(), n, static_cast<uint32_t>(opt.features_version), out.data(), status.data()));
    std::vector<BlissResult<Analysis>> res;
    res.reserve(n);
    for (uint32_t i = 0; i < n; i++) {
        if (status[i] == BLISSGPU_SONG_OK) res.emplace_back(Analysis(std::vector<float>(out.begin() + i * d, out.begin() + (i + 1) * d), opt.features_version));
        else res.emplace_back(AnalysisError("empty or too short song."));
    }
    return res;
}

// ---- Song (src/song/mod.rs:45-76, 373-522) ----
struct Song {
    std::string path;
    std::optional<std::string> artist, title, album, album_artist, genre;
    std::optional<int32_t> track_number, disc_number;
    double duration = 0.0;
    Analysis analysis;
    FeaturesVersion features_version = LATEST;

    static Analysis analyze(const std::vector<float>& sample_array) { return analyze_with_options(sample_array, AnalysisOptions{}); }
    // One song through blissgpu_analyze: safe from any number of threads (the reference's worker pool,
    // src/song/decoder.rs:299-329); concurrent calls are coalesced into one device batch inside the library.
    static Analysis analyze_with_options(const std::vector<float>& sample_array, const AnalysisOptions& opt) {
        std::vector<float> out(feature_count(opt.features_version));
        int32_t status = 0;
        const float dummy = 0.0f;
        check(blissgpu_analyze(sample_array.empty() ? &dummy : sample_array.data(), sample_array.size(),
                               static_cast<uint32_t>(opt.features_version), out.data(), &status));
        if (status != BLISSGPU_SONG_OK) throw AnalysisError("empty or too short song.");  // Err(...), src/song/mod.rs:426-430
        return Analysis(std::move(out), opt.features_version);
    }
    // Raw decoder output: `channels` interleaved channels of f32 (or s16, overload below) at 22 050 Hz; the mono downmix
    // ((L + R) * SQRT_2 / 2 for stereo, src/song/decoder/symphonia.rs:266-300) runs on the device.
    static Analysis analyze_interleaved(const std::vector<float>& samples, uint32_t channels, const AnalysisOptions& opt = {}) {
        return analyze_raw(samples.data(), BLISSGPU_SAMPLE_F32, channels, samples.size() / (channels ? channels : 1), opt);
    }
    static Analysis analyze_interleaved(const std::vector<int16_t>& samples, uint32_t channels, const AnalysisOptions& opt = {}) {
        return analyze_raw(samples.data(), BLISSGPU_SAMPLE_S16, channels, samples.size() / (channels ? channels : 1), opt);
    }
    static Analysis analyze_raw(const void* pcm, int sample_format, uint32_t channels, uint64_t frames, const AnalysisOptions& opt) {
        std::vector<float> out(feature_count(opt.features_version));
        int32_t status = 0;
        const float dummy = 0.0f;
        check(blissgpu_analyze_interleaved(frames ? pcm : &dummy, sample_format, channels, frames,
                                           static_cast<uint32_t>(opt.features_version), out.data(), &status));
        if (status != BLISSGPU_SONG_OK) throw AnalysisError("empty or too short song.");
        return Analysis(std::move(out), opt.features_version);
    }
    // Decoder output at ANY sample rate (`channels` interleaved channels of f32 / s16 / s32): what FFmpegDecoder::decode does
    // before Song::analyze -- libswresample to mono 22 050 Hz, src/song/decoder/ffmpeg.rs:36-109 -- runs on the device,
    // bit for bit (pinned by the reference's Adler-32 decoder tests, ffmpeg.rs:433-452).
    static Analysis analyze_decoded(const std::vector<float>& samples, uint32_t channels, uint32_t sample_rate, const AnalysisOptions& opt = {}) {
        return analyze_decoded_raw(samples.data(), BLISSGPU_SAMPLE_F32, channels, samples.size() / (channels ? channels : 1), sample_rate, opt);
    }
    static Analysis analyze_decoded(const std::vector<int16_t>& samples, uint32_t channels, uint32_t sample_rate, const AnalysisOptions& opt = {}) {
        return analyze_decoded_raw(samples.data(), BLISSGPU_SAMPLE_S16, channels, samples.size() / (channels ? channels : 1), sample_rate, opt);
    }
    static Analysis analyze_decoded(const std::vector<int32_t>& samples, uint32_t channels, uint32_t sample_rate, const AnalysisOptions& opt = {}) {
        return analyze_decoded_raw(samples.data(), BLISSGPU_SAMPLE_S32, channels, samples.size() / (channels ? channels : 1), sample_rate, opt);
    }
    static Analysis analyze_decoded_raw(const void* pcm, int sample_format, uint32_t channels, uint64_t frames, uint32_t sample_rate,
                                        const AnalysisOptions& opt) {
        std::vector<float> out(feature_count(opt.features_version));
        int32_t status = 0;
        const float dummy = 0.0f;
        check(blissgpu_analyze_decoded(frames ? pcm : &dummy, sample_format, channels, frames, sample_rate,
                                       static_cast<uint32_t>(opt.features_version), out.data(), &status));
        if (status != BLISSGPU_SONG_OK) throw AnalysisError("empty or too short song.");
        return Analysis(std::move(out), opt.features_version);
    }
    float distance(const Song& other) const { return analysis.distance(other.analysis); }
};

// Bulk form for a library as it comes off the decoders: every song with its own buffer, format, channel count and rate
// (blissgpu_analyze_batch_decoded); the compute half of analyze_paths_with_options (src/song/decoder.rs:278-332) for a host
// that keeps its decoder and drops the resampler.
inline std::vector<BlissResult<Analysis>> analyze_decoded_batch(const std::vector<blissgpu_decoded_song>& songs, const AnalysisOptions& opt = {}) {
    const uint32_t n = static_cast<uint32_t>(songs.size());
    const size_t d = feature_count(opt.features_version);
    std::vector<float> out(n * d);
    std::vector<int32_t> status(n);
    if (n) check(blissgpu_analyze_batch_decoded(songs.data(), n, static_cast<uint32_t>(opt.features_version), out.data(), status.data()));
    std::vector<BlissResult<Analysis>> res;
    res.reserve(n);
    for (uint32_t i = 0; i < n; i++) {
        if (status[i] == BLISSGPU_SONG_OK) res.emplace_back(Analysis(std::vector<float>(out.begin() + i * d, out.begin() + (i + 1) * d), opt.features_version));
        else res.emplace_back(AnalysisError("empty or too short song."));
    }
    return res;
}

// ---- CUE tracks as slices of one decoded buffer: BlissCueFile::get_songs (src/cue.rs:205-246) ----
// An INDEX time as the `std::time::Duration` the sheet parser (rcue in the reference) hands to BlissCueFile.
struct CueIndex {
    uint64_t secs;
    uint32_t nanos;
    // `Duration::as_secs_f32`: (secs as f32) + (nanos as f32) / 1e9_f32 -- NOT the exact sum rounded once: the two differ in the
    // last bit for 72 of the 360 000 mm:ss:ff values of an 80-minute disc, and the sample index derived from it moves by one
    float as_secs_f32() const {
        const volatile float q = static_cast<float>(nanos) / 1e9f;  // (rounded before the add, whatever the contraction mode)
        return static_cast<float>(secs) + q;
    }
    // mm:ss:ff of the sheet (75 frames per second), integer nanoseconds truncated
    static CueIndex from_msf(uint64_t mm, uint64_t ss, uint64_t ff) { return {mm * 60 + ss, static_cast<uint32_t>(ff * 1000000000ull / 75)}; }
};
// (start, end) sample ranges of the tracks of one FILE entry: `(index.as_secs_f32() * SAMPLE_RATE as f32) as usize` for each
// track's first INDEX (src/cue.rs:214-215,232); the last track runs to the end of the decoded file.  index_seconds must hold
// `Duration::as_secs_f32()` values (a float computed as mm * 60 + ss + ff / 75 is not the same number): use the CueIndex form
inline std::vector<std::pair<uint64_t, uint64_t>> cue_track_bounds(const std::vector<float>& index_seconds, uint64_t n_samples) {
    std::vector<std::pair<uint64_t, uint64_t>> b;
    for (size_t i = 0; i < index_seconds.size(); i++) {
        const uint64_t s = (uint64_t)(index_seconds[i] * (float)BLISSGPU_SAMPLE_RATE);
        const uint64_t e = i + 1 < index_seconds.size() ? (uint64_t)(index_seconds[i + 1] * (float)BLISSGPU_SAMPLE_RATE) : n_samples;
        b.emplace_back(s, e);
    }
    return b;
}
inline std::vector<std::pair<uint64_t, uint64_t>> cue_track_bounds(const std::vector<CueIndex>& index, uint64_t n_samples) {
    std::vector<float> secs;
    for (const CueIndex& ix : index) secs.push_back(ix.as_secs_f32());
    return cue_track_bounds(secs, n_samples);
}
// The audio file of a CUE sheet as the decoder delivered it (`frames` frames of `channels` interleaved samples at `sample_rate`) ->
// one result per track: the file is converted ONCE to mono 22 050 Hz on the device (FFmpegDecoder's conversion), the tracks are
// (offset, length) slices of that device buffer.  Sheet parsing (the reference uses the rcue crate) stays with the host.
// ctx: the context to run on; NULL = the library's first default context (borrowed: the one the entry points without a context
// argument use -- no context is created or destroyed per CUE file).
inline std::vector<BlissResult<Analysis>> analyze_cue_tracks(const void* pcm, int sample_format, uint32_t channels, uint64_t frames,
                                                             uint32_t sample_rate, const std::vector<float>& index_seconds,
                                                             const AnalysisOptions& opt = {}, blissgpu_ctx* ctx = nullptr) {
    struct Dev {  // device scratch of this call
        blissgpu_ctx* ctx = nullptr;
        void *raw = nullptr, *mono = nullptr, *rows = nullptr, *status = nullptr;
        ~Dev() { blissgpu_free(raw); blissgpu_free(mono); blissgpu_free(rows); blissgpu_free(status); }
    } d;
    if (channels == 0 || channels > 8) throw DecodingError("channels must be 1..8");
    if (sample_format != BLISSGPU_SAMPLE_F32 && sample_format != BLISSGPU_SAMPLE_S16 && sample_format != BLISSGPU_SAMPLE_S32)
        throw DecodingError("sample_format must be F32, S16 or S32");
    d.ctx = ctx;
    if (!d.ctx) check(blissgpu_default_ctx(0, &d.ctx));
    const size_t sample_bytes = sample_format == BLISSGPU_SAMPLE_S16 ? 2 : 4;
    const uint64_t n_mono = blissgpu_resampled_len(frames, sample_rate);
    const auto bounds = cue_track_bounds(index_seconds, n_mono);
    const uint32_t n = static_cast<uint32_t>(bounds.size());
    const size_t dft = feature_count(opt.features_version);
    std::vector<uint64_t> off(n), len(n);
    for (uint32_t i = 0; i < n; i++) {
        if (bounds[i].first > bounds[i].second || bounds[i].second > n_mono) throw DecodingError("CUE index beyond the end of the audio file");
        off[i] = bounds[i].first;
        len[i] = bounds[i].second - bounds[i].first;
    }
    check(blissgpu_malloc(&d.raw, std::max<uint64_t>(1, frames * channels * sample_bytes)));
    check(blissgpu_malloc(&d.mono, std::max<uint64_t>(1, n_mono) * 4 + 256));
    check(blissgpu_malloc(&d.rows, std::max<size_t>(1, n * dft) * 4));
    check(blissgpu_malloc(&d.status, std::max<uint32_t>(1, n) * 4));
    if (frames) check(blissgpu_memcpy_h2d(d.ctx, d.raw, pcm, frames * channels * sample_bytes));
    check(blissgpu_pcm_decode_device(d.ctx, d.raw, sample_format, channels, frames, sample_rate, static_cast<float*>(d.mono)));
    std::vector<float> out(n * dft);
    std::vector<int32_t> status(n);
    if (n) {
        check(blissgpu_analyze_batch_device(d.ctx, static_cast<const float*>(d.mono), off.data(), len.data(), n,
                                            static_cast<uint32_t>(opt.features_version), static_cast<float*>(d.rows),
                                            static_cast<int32_t*>(d.status)));
        check(blissgpu_memcpy_d2h(d.ctx, out.data(), d.rows, out.size() * 4));
        check(blissgpu_memcpy_d2h(d.ctx, status.data(), d.status, status.size() * 4));
    }
    std::vector<BlissResult<Analysis>> res;
    res.reserve(n);
    for (uint32_t i = 0; i < n; i++) {
        if (status[i] == BLISSGPU_SONG_OK) res.emplace_back(Analysis(std::vector<float>(out.begin() + i * dft, out.begin() + (i + 1) * dft), opt.features_version));
        else res.emplace_back(AnalysisError("empty or too short song."));
    }
    return res;
}
inline std::vector<BlissResult<Analysis>> analyze_cue_tracks(const void* pcm, int sample_format, uint32_t channels, uint64_t frames,
                                                             uint32_t sample_rate, const std::vector<CueIndex>& index,
                                                             const AnalysisOptions& opt = {}, blissgpu_ctx* ctx = nullptr) {
    std::vector<float> secs;
    for (const CueIndex& ix : index) secs.push_back(ix.as_secs_f32());
    return analyze_cue_tracks(pcm, sample_format, channels, frames, sample_rate, secs, opt, ctx);
}

// ---- Decoder trait (src/song/decoder.rs:34-333) ----
struct PreAnalyzedSong {
    std::string path;
    std::optional<std::string> artist, title, album, album_artist, genre;
    std::optional<int32_t> track_number, disc_number;
    double duration = 0.0;
    std::vector<float> sample_array;
    Song to_song(Analysis a, FeaturesVersion v) const {
        Song s;
        s.path = path; s.artist = artist; s.title = title; s.album = album; s.album_artist = album_artist; s.genre = genre;
        s.track_number = track_number; s.disc_number = disc_number; s.duration = duration;
        s.analysis = std::move(a); s.features_version = v;
        return s;
    }
};

class Decoder {
  public:
    virtual ~Decoder() = default;
    // required method (src/song/decoder.rs:129): file -> mono 22 050 Hz f32 samples; throws BlissError
    virtual PreAnalyzedSong decode(const std::string& path) const = 0;

    Song song_from_path(const std::string& path) const { return song_from_path_with_options(path, AnalysisOptions{}); }
    Song song_from_path_with_options(const std::string& path, const AnalysisOptions& opt) const {
        PreAnalyzedSong pre = decode(path);
        return pre.to_song(Song::analyze_with_options(pre.sample_array, opt), opt.features_version);
    }
    // (path, Result) pairs; a bad file never aborts the run (src/song/decoder.rs:313-325)
    std::vector<std::pair<std::string, BlissResult<Song>>> analyze_paths(const std::vector<std::string>& paths, const AnalysisOptions& opt = {}) const {
        std::vector<std::pair<std::string, BlissResult<Song>>> out;
        std::vector<PreAnalyzedSong> pending;
        for (const auto& p : paths) {
            try { pending.push_back(decode(p)); }
            catch (const BlissError& e) { out.emplace_back(p, e); }
        }
        std::vector<std::vector<float>> pcm;
        pcm.reserve(pending.size());
        for (auto& s : pending) pcm.push_back(s.sample_array);
        auto res = analyze_batch(pcm, opt);
        for (size_t i = 0; i < pending.size(); i++) {
            if (auto* e = std::get_if<BlissError>(&res[i])) out.emplace_back(pending[i].path, *e);
            else out.emplace_back(pending[i].path, pending[i].to_song(std::get<Analysis>(std::move(res[i])), opt.features_version));
        }
        return out;
    }
};

// ---- playlist ordering (src/playlist.rs:24-59, 173-221, 256-402); distances are evaluated on the GPU ----
// A metric builder names one of the device metrics (the reference's `&dyn DistanceMetricBuilder` implemented by the
// plain distance functions, src/playlist.rs:41-59); the metric it builds from a set of vectors is the sum of the
// distances to each vector of the set.
struct MetricBuilder {
    int metric = BLISSGPU_METRIC_EUCLIDEAN;
    std::vector<float> m;  // d x d row-major for MAHALANOBIS
    const float* mptr() const { return m.empty() ? nullptr : m.data(); }
};
inline MetricBuilder euclidean_builder() { return {BLISSGPU_METRIC_EUCLIDEAN, {}}; }
inline MetricBuilder cosine_builder() { return {BLISSGPU_METRIC_COSINE, {}}; }
inline MetricBuilder mahalanobis_builder(std::vector<float> m) { return {BLISSGPU_METRIC_MAHALANOBIS, std::move(m)}; }

inline void check_ordering(int rc) {  // n32() / argmin().unwrap() panic on NaN in the reference
    if (rc == BLISSGPU_ERR_NAN) throw std::domain_error("NaN distance");
    check(rc);
}

template <typename T>
const Song& as_song(const T& s) { return s; }          // AsRef<Song>: specialise for wrapper types
template <typename T>
const Song& as_song(T* const& s) { return as_song(*s); }

template <typename T>
std::vector<float> feature_matrix(const std::vector<T>& songs, size_t& d) {
    std::vector<float> x;
    d = songs.empty() ? 0 : as_song(songs[0]).analysis.as_vec().size();
    for (const auto& s : songs) {
        const auto& v = as_song(s).analysis.as_vec();
        if (v.size() != d) throw std::logic_error("Mismatched features version between two songs or analysis");
        x.insert(x.end(), v.begin(), v.end());
    }
    return x;
}

// closest_to_songs (src/playlist.rs:256-270)
template <typename T>
std::vector<T> closest_to_songs(const std::vector<T>& initial_songs, const std::vector<T>& candidate_songs, const MetricBuilder& mb) {
    if (candidate_songs.empty()) return {};
    size_t d = 0, ds = 0;
    const auto x = feature_matrix(candidate_songs, d);
    const auto s = feature_matrix(initial_songs, ds);
    std::vector<uint32_t> order(candidate_songs.size());
    check_ordering(blissgpu_closest_to_songs(s.data(), (uint32_t)initial_songs.size(), x.data(), candidate_songs.size(), (uint32_t)d,
                                             mb.metric, mb.mptr(), order.data(), nullptr));
    std::vector<T> out;
    for (uint32_t i : order) out.push_back(candidate_songs[i]);
    return out;
}

// song_to_song (src/playlist.rs:272-326)
template <typename T>
std::vector<T> song_to_song(const std::vector<T>& initial_songs, const std::vector<T>& candidate_songs, const MetricBuilder& mb) {
    if (candidate_songs.empty()) return {};
    size_t d = 0, ds = 0;
    const auto x = feature_matrix(candidate_songs, d);
    const auto s = feature_matrix(initial_songs, ds);
    std::vector<uint32_t> order(candidate_songs.size());
    check_ordering(blissgpu_song_to_song(s.data(), (uint32_t)initial_songs.size(), x.data(), candidate_songs.size(), (uint32_t)d,
                                         mb.metric, mb.mptr(), order.data()));
    std::vector<T> out;
    for (uint32_t i : order) out.push_back(candidate_songs[i]);
    return out;
}

// dedup_playlist_custom_distance / dedup_playlist (src/playlist.rs:343-402)
template <typename T>
std::vector<T> dedup_playlist_custom_distance(const std::vector<T>& playlist, std::optional<float> distance_threshold, const MetricBuilder& mb) {
    const float thr = distance_threshold.value_or(0.05f);
    size_t d = 0;
    const auto x = feature_matrix(playlist, d);
    const size_t n = playlist.size(), window = 64;
    std::vector<T> out;
    std::vector<float> dist(window);
    size_t i = 0;
    while (i < n) {
        size_t j = i + 1;
        bool stopped = false;
        while (j < n && !stopped) {
            const size_t hi = std::min(n, j + window);
            check(blissgpu_set_distance(x.data() + i * d, 1, x.data() + j * d, hi - j, (uint32_t)d, mb.metric, mb.mptr(), dist.data()));
            size_t k = j;
            for (; k < hi; k++) {
                const float dk = dist[k - j];
                if (dk != dk) throw std::domain_error("NaN distance");
                const Song &a = as_song(playlist[i]), &b = as_song(playlist[k]);
                const bool same = dk < thr || (a.title && b.title && a.artist && b.artist && *a.title == *b.title && *a.artist == *b.artist);
                if (!same) { stopped = true; break; }
            }
            j = k;
        }
        out.push_back(playlist[i]);
        i = j;
    }
    return out;
}
template <typename T>
std::vector<T> dedup_playlist(const std::vector<T>& playlist, std::optional<float> distance_threshold) {
    return dedup_playlist_custom_distance(playlist, distance_threshold, euclidean_builder());
}

// closest_album_to_group (src/playlist.rs:424-485): the albums of `pool` (songs of `group` removed, songs without an
// album dropped) ordered by the euclidean distance of their mean analysis to the group's mean analysis, each album
// ordered by (disc number, track number); `group` itself comes first.
template <typename T>
std::vector<T> closest_album_to_group(const std::vector<T>& group, const std::vector<T>& pool_in) {
    if (group.empty()) throw ProviderError("Mean of empty slice");
    std::vector<T> pool;
    for (const auto& s : pool_in) {
        bool in_group = false;
        for (const auto& g : group) {
            const Song &a = as_song(g), &b = as_song(s);
            if (a.path == b.path && a.analysis == b.analysis && a.album == b.album && a.title == b.title && a.artist == b.artist &&
                a.track_number == b.track_number && a.disc_number == b.disc_number) { in_group = true; break; }
        }
        if (!in_group) pool.push_back(s);
    }
    const size_t d = as_song(group[0]).analysis.as_vec().size();
    auto mean_of = [&](const std::vector<const Song*>& songs) {  // ndarray mean_axis: sequential f32 row sum / n
        std::vector<float> m(d, 0.0f);
        for (const Song* s : songs) for (size_t k = 0; k < d; k++) m[k] = m[k] + s->analysis.as_vec()[k];
        for (size_t k = 0; k < d; k++) m[k] = m[k] / (float)songs.size();
        return m;
    };
    std::vector<std::string> names;
    std::vector<std::vector<const Song*>> members;
    for (const auto& s : pool) {
        const Song& song = as_song(s);
        if (!song.album) continue;
        size_t a = 0;
        while (a < names.size() && names[a] != *song.album) a++;
        if (a == names.size()) { names.push_back(*song.album); members.emplace_back(); }
        members[a].push_back(&song);
    }
    std::vector<const Song*> gs;
    for (const auto& g : group) gs.push_back(&as_song(g));
    const std::vector<float> first = mean_of(gs);
    std::vector<T> playlist = group;
    if (!names.empty()) {
        std::vector<float> means;
        for (const auto& m : members) { const auto v = mean_of(m); means.insert(means.end(), v.begin(), v.end()); }
        std::vector<uint32_t> order(names.size());
        check_ordering(blissgpu_closest_to_songs(first.data(), 1, means.data(), names.size(), (uint32_t)d, BLISSGPU_METRIC_EUCLIDEAN,
                                                 nullptr, order.data(), nullptr));
        for (uint32_t a : order) {
            std::vector<T> al;
            for (const auto& s : pool) if (as_song(s).album && *as_song(s).album == names[a]) al.push_back(s);
            std::stable_sort(al.begin(), al.end(), [](const T& x, const T& y) {  // Option<i32>: None < Some
                const Song &p = as_song(x), &q = as_song(y);
                return std::make_pair(p.disc_number, p.track_number) < std::make_pair(q.disc_number, q.track_number);
            });
            playlist.insert(playlist.end(), al.begin(), al.end());
        }
    }
    return playlist;
}

// variance_based_weight_matrix (src/playlist.rs:173-221): d x d row-major; host arithmetic in the reference's order
inline std::vector<float> variance_based_weight_matrix(const std::vector<std::vector<float>>& seeds) {
    if (seeds.size() < 2) throw ProviderError("seeds must contain more than one element");
    const size_t n = seeds[0].size();
    if (n == 0) throw ProviderError("seed feature vectors must not be empty");
    for (const auto& s : seeds) if (s.size() != n) throw ProviderError("all seed feature vectors must have the same length");
    const float ns = (float)seeds.size();
    std::vector<float> mean(n, 0.0f), var(n, 0.0f);
    for (const auto& s : seeds) for (size_t k = 0; k < n; k++) mean[k] = mean[k] + s[k];
    for (size_t k = 0; k < n; k++) mean[k] = mean[k] / ns;
    for (const auto& s : seeds) for (size_t k = 0; k < n; k++) { const float diff = s[k] - mean[k]; var[k] = var[k] + diff * diff; }
    for (size_t k = 0; k < n; k++) { var[k] = var[k] / ns; var[k] = 1.0f / (var[k] + 1e-6f); }
    float p[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sum = 0.0f;  // ndarray sum(): unrolled_fold
    size_t k = 0;
    for (; k + 8 <= n; k += 8) for (int u = 0; u < 8; u++) p[u] = p[u] + var[k + u];
    sum = sum + (p[0] + p[4]); sum = sum + (p[1] + p[5]); sum = sum + (p[2] + p[6]); sum = sum + (p[3] + p[7]);
    for (; k < n; k++) sum = sum + var[k];
    const float scale = (float)n / sum;
    std::vector<float> m(n * n, 0.0f);
    for (size_t q = 0; q < n; q++) m[q * n + q] = var[q] * scale;
    return m;
}

}  // namespace bliss
