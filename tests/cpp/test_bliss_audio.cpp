// C++ host-mirror test: reads like the reference's own unit tests (src/song/mod.rs:539-633,
// src/playlist.rs:1008-1110, src/lib.rs:272-291) but runs on the GPU through libblissgpu.so.
//   usage: test_bliss_audio <golden pcm s16 raw file> <expected 23 floats file> [<stereo s16 raw> [<44.1 kHz stereo s32 raw> <its 23 floats>]]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>

#include "../../bliss-rs_amd/csrc/bliss_audio.hpp"

using namespace bliss;

#define CHECK(cond) do { if (!(cond)) { std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); std::exit(1); } } while (0)

struct RawS16Decoder : Decoder {  // ffmpeg's s16 -> flt conversion: sample / 32768
    PreAnalyzedSong decode(const std::string& path) const override {
        std::ifstream f(path, std::ios::binary);
        if (!f) throw DecodingError("while opening format for file '" + path + "'");
        std::vector<char> raw((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        PreAnalyzedSong s;
        s.path = path;
        const int16_t* p = reinterpret_cast<const int16_t*>(raw.data());
        s.sample_array.resize(raw.size() / 2);
        for (size_t i = 0; i < s.sample_array.size(); i++) s.sample_array[i] = (float)p[i] / 32768.0f;
        s.duration = (double)s.sample_array.size() / SAMPLE_RATE;
        return s;
    }
};

int main(int argc, char** argv) {
    CHECK(argc == 3 || argc == 4 || argc == 6 || argc == 8);
    // test_analysis_too_small (src/song/mod.rs:539-551)
    try { Song::analyze({0.0f}); CHECK(false); } catch (const BlissError& e) { CHECK(e == AnalysisError("empty or too short song.")); }
    try { Song::analyze({}); CHECK(false); } catch (const BlissError& e) { CHECK(e == AnalysisError("empty or too short song.")); }
    // test_analyze (src/song/mod.rs:553-591), tolerance 1e-5 like the reference
    std::vector<float> expected(23);
    { std::ifstream f(argv[2]); for (auto& v : expected) f >> v; }
    RawS16Decoder dec;
    Song song = dec.song_from_path(argv[1]);
    for (size_t i = 0; i < 23; i++) CHECK(std::fabs(song.analysis.as_vec()[i] - expected[i]) < 1e-5f);
    CHECK(std::fabs(song.analysis[AnalysisIndex::Zcr] - (-0.849141f)) < 1e-6f);
    AnalysisOptions v1{FeaturesVersion::Version1, 1};
    CHECK(dec.song_from_path_with_options(argv[1], v1).analysis.as_vec().size() == 20);
    // s16 feed: widened on the device like FFmpeg's s16 -> flt conversion => bit-identical analysis
    {
        std::ifstream f(argv[1], std::ios::binary);
        std::vector<char> raw((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        std::vector<int16_t> s16(raw.size() / 2);
        std::memcpy(s16.data(), raw.data(), s16.size() * 2);
        auto r = analyze_batch(std::vector<std::vector<int16_t>>{s16, std::vector<int16_t>(10, 0)});
        CHECK(std::get<Analysis>(r[0]) == song.analysis);
        CHECK(std::get<BlissError>(r[1]) == AnalysisError("empty or too short song."));
    }
    // stereo decoder output: the mono downmix runs on the device ((L + R) * SQRT_2 / 2, src/song/decoder/symphonia.rs:281-285)
    if (argc >= 4) {
        std::ifstream f(argv[3], std::ios::binary);
        std::vector<char> raw((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        std::vector<int16_t> st(raw.size() / 2);
        std::memcpy(st.data(), raw.data(), st.size() * 2);
        std::vector<float> mono(st.size() / 2);
        for (size_t i = 0; i < mono.size(); i++) {
            const float l = (float)st[2 * i] / 32768.0f, r = (float)st[2 * i + 1] / 32768.0f;
            volatile float sum = l + r;               // every step rounds to f32, like the Rust expression
            volatile float scaled = sum * 1.41421356237309504880f;
            mono[i] = scaled / 2.0f;
        }
        CHECK(Song::analyze_interleaved(st, 2) == Song::analyze(mono));
    }
    // decoder output at another rate (44.1 kHz stereo, 24-bit samples in s32): FFmpegDecoder's conversion runs on the device
    // (src/song/decoder/ffmpeg.rs:36-109); argv[5] holds the row the Python mirror got from the same call
    if (argc >= 6) {
        std::ifstream f(argv[4], std::ios::binary);
        std::vector<char> raw((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        std::vector<int32_t> st(raw.size() / 4);
        std::memcpy(st.data(), raw.data(), st.size() * 4);
        std::vector<float> want(23);
        { std::ifstream g(argv[5]); for (auto& v : want) g >> v; }
        const Analysis got = Song::analyze_decoded(st, 2, 44100);
        for (size_t i = 0; i < 23; i++) CHECK(got.as_vec()[i] == want[i]);
        CHECK(blissgpu_resampled_len(st.size() / 2, 44100) == (st.size() / 2 + 1) / 2);
        // the bulk form with two different songs of the same file: the row above, and a too-short one
        std::vector<blissgpu_decoded_song> lib = {{st.data(), st.size() / 2, 44100, 2, BLISSGPU_SAMPLE_S32},
                                                  {st.data(), 1000, 48000, 2, BLISSGPU_SAMPLE_S32}};
        auto r = analyze_decoded_batch(lib);
        CHECK(std::get<Analysis>(r[0]) == got);
        CHECK(std::get<BlissError>(r[1]) == AnalysisError("empty or too short song."));
    }
    // BlissCue::songs_from_path on data/testcue.cue (src/cue.rs:270-415): argv[6] = the 44.1 kHz stereo s16 frames of testcue.flac,
    // argv[7] = the 3 x 23 literals the reference asserts; the tracks are slices of ONE device-side conversion
    if (argc == 8) {
        std::ifstream f(argv[6], std::ios::binary);
        std::vector<char> raw((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        std::vector<int16_t> st(raw.size() / 2);
        std::memcpy(st.data(), raw.data(), st.size() * 2);
        std::vector<float> want(69);
        { std::ifstream g(argv[7]); for (auto& v : want) g >> v; }
        const std::vector<CueIndex> idx = {CueIndex::from_msf(0, 0, 0), CueIndex::from_msf(0, 11, 5), CueIndex::from_msf(0, 16, 69)};  // INDEX 01 of data/testcue.cue
        auto tracks = analyze_cue_tracks(st.data(), BLISSGPU_SAMPLE_S16, 2, st.size() / 2, 44100, idx);
        CHECK(tracks.size() == 3);
        for (size_t t = 0; t < 3; t++) {
            const auto& a = std::get<Analysis>(tracks[t]).as_vec();
            for (size_t i = 0; i < 23; i++) CHECK(std::fabs(a[i] - want[23 * t + i]) < 1e-5f);
        }
        const auto b = cue_track_bounds(idx, 496272);
        CHECK(b[0].second == 244020 && b[1].first == 244020 && b[1].second == 373086 && b[2].second == 496272);
        // Duration::as_secs_f32 is secs as f32 + nanos as f32 / 1e9, not the exact sum rounded once: 00:00:05 starts at sample
        // 1469 (the single rounding says 1470), 00:00:11 at 3234 (3233) -- src/cue.rs:214-215
        const auto o = cue_track_bounds(std::vector<CueIndex>{CueIndex::from_msf(0, 0, 5), CueIndex::from_msf(0, 0, 11), CueIndex::from_msf(0, 1, 14)}, 99999);
        CHECK(o[0].first == 1469 && o[1].first == 3234 && o[2].first == 26166);
        CHECK((uint64_t)((0.0f + 5.0f / 75.0f) * 22050.0f) == 1470);
    }
    // bulk path: a missing file is reported, not fatal (src/song/decoder.rs:313-325)
    auto res = dec.analyze_paths({argv[1], "/nonexistent.raw"});
    CHECK(res.size() == 2);
    int ok = 0, bad = 0;
    for (auto& [p, r] : res) { if (std::holds_alternative<Song>(r)) { ok++; CHECK(std::get<Song>(r).analysis == song.analysis); } else { bad++; CHECK(std::get<BlissError>(r).kind == BlissError::Kind::DecodingError); } }
    CHECK(ok == 1 && bad == 1);
    // distances (assert_eq on f32 in the reference)
    std::vector<float> a(20, 1.0f), b(20, 0.0f);
    a[19] = 0.0f; b[16] = 1.0f;
    CHECK(euclidean_distance(a, b) == 4.2426405f);
    CHECK(cosine_distance(a, b) == 0.7705842661294382f);
    CHECK(distance_metric(FeaturesVersion::Version1)(std::vector<float>(20, 0.f), std::vector<float>(20, 1.f)) == 4.47213595f);
    CHECK(distance_metric(FeaturesVersion::Version2)(std::vector<float>(23, 0.f), std::vector<float>(23, 1.f)) == 3.4999998f);
    Analysis z(std::vector<float>(20, 0.f), FeaturesVersion::Version1), o(std::vector<float>(20, 1.f), FeaturesVersion::Version1);
    CHECK(z.distance(o) == 4.472136f);
    try { z.distance(Analysis(std::vector<float>(23, 0.f), FeaturesVersion::Version2)); CHECK(false); } catch (const std::logic_error&) {}
    try { Analysis(std::vector<float>(3, 0.f), LATEST); CHECK(false); } catch (const BlissError& e) { CHECK(e.kind == BlissError::Kind::ProviderError); }
    try { features_version_try_from(3); CHECK(false); } catch (const BlissError& e) { CHECK(e.message == "This features' version (3) does not exist"); }
    // ---- playlist ordering: the reference's own cases (src/playlist.rs:506-1007), pointers stand in for &Song ----
    {
        auto mk = [](const char* path, std::vector<float> v, const char* title = nullptr, const char* artist = nullptr) {
            Song s2;
            s2.path = path;
            s2.analysis = Analysis(std::move(v), LATEST);
            if (title) s2.title = title;
            if (artist) s2.artist = artist;
            return s2;
        };
        auto vec16 = [](float head, float x16) { std::vector<float> v(23, 1.0f); for (int i = 0; i < 16; i++) v[i] = head; v[16] = x16; return v; };
        Song first = mk("path-to-first", std::vector<float>(23, 1.0f)), dupe = mk("path-to-dupe", std::vector<float>(23, 1.0f));
        Song second = mk("path-to-second", vec16(2.0f, 1.9f), "dupe-title", "dupe-artist");
        Song third = mk("path-to-third", vec16(2.0f, 2.5f), "dupe-title", "dupe-artist");
        Song fourth = mk("path-to-fourth", vec16(2.0f, 0.0f), "dupe-title", "no-dupe-artist");
        Song fifth = mk("path-to-fourth", vec16(2.0f, 0.001f));
        Song fifth_b = mk("path-to-fifth", vec16(2.0f, 0.0f));
        using V = std::vector<const Song*>;
        // test_song_to_song (:733-858)
        CHECK((song_to_song(V{&first}, V{&first, &third, &dupe, &second, &fourth}, euclidean_builder()) == V{&first, &dupe, &second, &third, &fourth}));
        CHECK((song_to_song(V{&first}, V{&first, &dupe, &third, &fourth, &second}, euclidean_builder()) == V{&first, &dupe, &second, &third, &fourth}));
        // test_sort_closest_to_songs (:860-1007): equal distances keep the candidates' order
        CHECK((closest_to_songs(V{&first}, V{&fifth_b, &fourth, &first, &dupe, &second, &third}, euclidean_builder()) == V{&first, &dupe, &second, &fifth_b, &fourth, &third}));
        CHECK((closest_to_songs(V{&first}, V{&second, &first, &fourth, &dupe, &third, &fifth_b}, euclidean_builder()) == V{&first, &dupe, &second, &fourth, &fifth_b, &third}));
        // test_dedup_playlist_custom_distance (:506-731)
        const V pl{&first, &dupe, &second, &third, &fourth, &fifth};
        CHECK((dedup_playlist_custom_distance(pl, std::nullopt, euclidean_builder()) == V{&first, &second, &fourth}));
        CHECK((dedup_playlist_custom_distance(pl, 20.0f, euclidean_builder()) == V{&first}));
        CHECK((dedup_playlist_custom_distance(pl, 20.0f, cosine_builder()) == V{&first}));
        CHECK((dedup_playlist(pl, 20.0f) == V{&first}));
        CHECK((dedup_playlist(pl, std::nullopt) == V{&first, &second, &fourth}));
        // test_closest_to_group (:1112-1260)
        {
            auto song = [](const char* path, const char* album, const char* artist, int track, int disc, float value) {
                Song s3;
                s3.path = path; s3.artist = artist; s3.track_number = track; s3.disc_number = disc;
                if (album) s3.album = album;
                s3.analysis = Analysis(std::vector<float>(23, value), LATEST);
                return s3;
            };
            Song a = song("path-to-first", "Album", "Artist", 1, 1, 0.0f), b = song("path-to-third", "Album", "Another Artist", 2, 1, 10.0f);
            Song o1 = song("path-to-second-2", "Another Album", "Artist", 1, 1, 0.15f), o2 = song("path-to-second", "Another Album", "Artist", 2, 1, 0.1f);
            Song p1 = song("path-to-fourth", "Another Album", "Another Artist", 1, 2, 20.0f), p2 = song("path-to-fourth", "Another Album", "Another Artist", 4, 2, 20.0f);
            Song none = song("path-to-fifth", nullptr, "Third Artist", 0, 0, 40.0f);
            none.track_number.reset(); none.disc_number.reset();
            const V got = closest_album_to_group(V{&a, &b}, V{&a, &o2, &p2, &b, &p1, &o1, &none});
            CHECK((got == V{&a, &b, &o1, &o2, &p1, &p2}));
        }
        // variance_based_weight_matrix (:1663-1765)
        auto m = variance_based_weight_matrix({{1.0f, 0.0f, 1.0f}, {1.0f, 100.0f, 1.0f}, {1.0f, 200.0f, 1.0f}});
        CHECK(m.size() == 9 && m[0] > m[4] && m[8] > m[4] && m[1] == 0.0f && m[3] == 0.0f);
        CHECK(std::fabs(m[0] + m[4] + m[8] - 3.0f) < 1e-4f);
        try { variance_based_weight_matrix({{1.0f, 2.0f, 3.0f}}); CHECK(false); } catch (const BlissError& e) { CHECK(e == ProviderError("seeds must contain more than one element")); }
        try { variance_based_weight_matrix({{1.0f, 2.0f, 3.0f}, {1.0f, 2.0f}}); CHECK(false); } catch (const BlissError& e) { CHECK(e == ProviderError("all seed feature vectors must have the same length")); }
        try { variance_based_weight_matrix({{}, {}}); CHECK(false); } catch (const BlissError& e) { CHECK(e == ProviderError("seed feature vectors must not be empty")); }
        // the adaptive metric plugs into the Mahalanobis mode of the ordering functions
        auto w = variance_based_weight_matrix({first.analysis.as_vec(), second.analysis.as_vec()});
        CHECK((closest_to_songs(V{&first}, V{&third, &first}, mahalanobis_builder(w)) == V{&first, &third}));
    }
    std::puts("test_bliss_audio: all checks passed");
    return 0;
}
