#!/usr/bin/env python3
"""Generate tests/golden/ from the reference's DATA files (run once, in the build container).

Nothing here executes or copies reference *source*: it copies the small `.npy` data fixtures the
reference's own unit tests load (data/*.npy) and six of its audio data files, decodes two more audio data files to raw PCM
with the test-tool FLAC decoder (tests/tools/flac_decode.py) and checks the decoded PCM against
the Adler-32 values the reference pins (src/song/decoder/ffmpeg.rs:455-462, :524-527).
The reference's known-answer literals (numbers asserted in its unit tests) are recorded in
reference_literals.json with the file:line they come from.

    python tests/golden/make_fixtures.py [/root/reference]
"""
import json
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "tools"))
from flac_decode import adler32_f32le, decode_flac  # noqa: E402

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
DATA = os.path.join(REF, "data")

NPY = [
    # name                         pinned by (reference test)                      tolerance
    "chroma-filter.npy",           # src/chroma.rs:704-714  chroma_filter(22050,2048,12,-0.1)   1e-9
    "chroma-interval.npy",         # src/chroma.rs:511-540  input of extract_interval_features
    "interval-feature-matrix.npy", # src/chroma.rs:511-540  expected output                      1e-7
    "chroma.npy",                  # src/chroma.rs:621-639  chroma_stft of the golden song      1e-7
    "pitch-tuning.npy",            # src/chroma.rs:667-673  pitch_tuning(.., 0.05, 12) == -0.1
    "spectrum-chroma.npy",         # src/chroma.rs:641-648, 681-702  estimate_tuning / pip_track input
    "spectrum-chroma-pitches.npy", # src/chroma.rs:681-702                                       1e-8
    "spectrum-chroma-mags.npy",    # src/chroma.rs:681-702                                       1e-8
    "librosa-decoded.npy",         # == decoded data/piano.flac (Adler-32 0xde831e82)
    "librosa-stft.npy",            # src/utils.rs:527-541   stft(piano, 2048, 512)               1e-4
]


# Audio DATA files of the reference's decoder / CUE / timbral tests that are NOT at 22 050 Hz: inputs of the resampler
# row.  Copied as they are (data, 3.3 MB together); the tests decode them with tests/tools/flac_decode.py
# / the stdlib wave module.  What the reference pins on each is in reference_literals.json ("resample").
AUDIO = [
    "s32_mono_44_1_kHz.flac",    # src/song/decoder/ffmpeg.rs:433-438  Adler-32 0xa0f8b8af after FFmpeg's conversion
    "s32_stereo_44_1_kHz.flac",  # :440-445  0xbbcba1cf
    "no_channel.wav",            # :471-476  0xd594429c (44.1 kHz mono s16)
    "testcue.flac",              # src/cue.rs:270-415  three tracks x 23 features (44.1 kHz stereo s16), with testcue.cue's indices
    "tone_11080Hz.flac",         # src/timbral.rs:364-373, 430-439  centroid / rolloff of an 11 080 Hz tone (44.1 kHz mono s16)
    "flush_test_52000.wav",      # src/song/decoder/symphonia.rs:713  48 kHz mono s16, 52 000 frames -> 23 888 samples
]


def main():
    for name in NPY:
        shutil.copyfile(os.path.join(DATA, name), os.path.join(HERE, name))
    for name in AUDIO:
        shutil.copyfile(os.path.join(DATA, name), os.path.join(HERE, name))
    # a library file as an older bliss-rs wrote it: the DATA file its own upgrade test loads
    # (src/library.rs:3935-4002, data/old_database.sql) -- rows + the old schema, no program text
    shutil.copyfile(os.path.join(DATA, "old_database.sql"), os.path.join(HERE, "old_database.sql"))

    # golden song -> s16 PCM (exactly what ffmpeg hands the reference, before the /32768 scaling)
    a, sr, bps = decode_flac(os.path.join(DATA, "s16_mono_22_5kHz.flac"))
    assert (sr, bps, a.shape[1]) == (22050, 16, 1)
    pcm = (a[:, 0] / 32768.0).astype(np.float32)
    assert adler32_f32le(pcm) == 0x5E01930B, hex(adler32_f32le(pcm))
    np.save(os.path.join(HERE, "s16_mono_22_5kHz.pcm_s16.npy"), a[:, 0].astype(np.int16))

    # piano.flac decodes to librosa-decoded.npy bit-for-bit (reference hash 0xde831e82)
    p, sr, bps = decode_flac(os.path.join(DATA, "piano.flac"))
    ppcm = (p[:, 0] / 32768.0).astype(np.float32)
    assert adler32_f32le(ppcm) == 0xDE831E82
    assert np.array_equal(ppcm, np.load(os.path.join(HERE, "librosa-decoded.npy")))

    # stereo 22.05 kHz twin of the golden song: kept as s16 [n,2] for the downmix ("next" row f1)
    s, sr, bps = decode_flac(os.path.join(DATA, "s16_stereo_22_5kHz.flac"))
    assert (sr, bps, s.shape[1]) == (22050, 16, 2)
    np.save(os.path.join(HERE, "s16_stereo_22_5kHz.pcm_s16.npy"), s.astype(np.int16))

    literals = {
        "_provenance": "numbers asserted by the reference's own unit tests; see 'src' of each entry",
        "analysis_v2_s16_mono_22_5kHz": {
            "src": "src/song/mod.rs:553-591", "tol": 1e-5,
            "values": [0.3846389, -0.849141, -0.75481045, -0.8790748, -0.63258266, -0.7258959,
                       -0.7757379, -0.8146726, 0.2716726, 0.25779057, -0.34292513, -0.62803423,
                       -0.28095096, 0.08686459, 0.24446082, -0.5723257, 0.23292065, 0.19981146,
                       -0.58594406, -0.06784296, -0.06000763, -0.58485717, -0.07880378]},
        "analysis_v1_s16_mono_22_5kHz": {
            "src": "src/song/mod.rs:593-633", "tol": 1e-5,
            "values": [0.3846389, -0.849141, -0.75481045, -0.8790748, -0.63258266, -0.7258959,
                       -0.7757379, -0.8146726, 0.2716726, 0.25779057, -0.35661936, -0.63578653,
                       -0.29593682, 0.06421304, 0.21852458, -0.581239, -0.9466835, -0.9481153,
                       -0.9820945, -0.95968974]},
        "chroma_desc_v2_first10": {
            "src": "src/chroma.rs:569-592", "tol": 1e-7,
            "values": [-0.34292513, -0.62803423, -0.28095096, 0.08686459, 0.24446082, -0.5723257,
                       0.23292065, 0.19981146, -0.58594406, -0.06784296]},
        "chroma_desc_v1": {
            "src": "src/chroma.rs:594-619", "tol": 1e-7,
            "values": [-0.35661936, -0.63578653, -0.29593682, 0.06421304, 0.21852458, -0.581239,
                       -0.9466835, -0.9481153, -0.9820945, -0.95968974]},
        "chroma_interval_features_of_chroma_npy": {
            "src": "src/chroma.rs:497-509", "tol": 1e-8,
            "values": [0.03860284, 0.02185281, 0.04224379, 0.06385278, 0.07311148, 0.02512566,
                       0.00319899, 0.00311308, 0.00107433, 0.00241861]},
        "estimate_tuning_spectrum_chroma": {"src": "src/chroma.rs:641-648", "tol": 1e-6,
                                            "value": -0.09999999999999998, "n_fft": 2048},
        "estimate_tuning_golden_song": {"src": "src/chroma.rs:655-665", "tol": 1e-6,
                                        "value": -0.04999999999999999},
        "pitch_tuning_fixture": {"src": "src/chroma.rs:667-673", "value": -0.1, "resolution": 0.05},
        "timbral_chunks_exact": {
            "_note": "these tests frame with chunks_exact(HOP) (src/timbral.rs:295-299), not analyze's windows(512).step_by(hop)",
            "zcr": {"src": "src/timbral.rs:290-299", "tol": 1e-3, "value": -0.85036},
            "flatness": {"src": "src/timbral.rs:334-349", "tol": 1e-2, "values": [-0.77610075, -0.8148179]},
            "rolloff": {"src": "src/timbral.rs:380-395", "tol": 1e-2, "values": [-0.6326486, -0.7260933]},
            "centroid": {"src": "src/timbral.rs:397-413", "tol": 1e-4, "values": [-0.75483, -0.87916887]}},
        "tempo_real_chunks_exact": {"src": "src/temporal.rs:100-108", "tol": 1e-2, "value": 0.378605},
        "tempo_artificial_60bpm": {"src": "src/temporal.rs:120-138", "tol": 1e-2, "value": -0.416853},
        "tempo_artificial_192bpm": {"src": "src/temporal.rs:140-161", "tol": 1e-2, "value": 0.86},
        "loudness_chunks_exact": {"src": "src/misc.rs:85-96", "tol": 1e-2, "values": [0.271263, 0.2577181]},
        "distances": {
            "euclidean": {"src": "src/playlist.rs:1081-1095", "value": 4.242640687119285},
            "cosine": {"src": "src/playlist.rs:1097-1110", "value": 0.7705842661294382},
            "mahalanobis": {"src": "src/playlist.rs:1008-1024", "value": 1.0},
            "v1_metric_zeros_ones": {"src": "src/lib.rs:272-281", "value": 4.47213595},
            "v2_metric_zeros_ones": {"src": "src/lib.rs:283-290", "value": 3.4999998}},
        "adler32": {"s16_mono_22_5kHz": "0x5e01930b", "piano": "0xde831e82",
                    "src": "src/song/decoder/ffmpeg.rs:455-462,524-527"},
        "resample": {
            "_note": "FFmpegDecoder's output (libswresample, default options) on files that are not at 22 050 Hz",
            "adler32": {"src": "src/song/decoder/ffmpeg.rs:433-452,471-476",
                        "s32_mono_44_1_kHz.flac": "0xa0f8b8af", "s32_stereo_44_1_kHz.flac": "0xbbcba1cf",
                        "no_channel.wav": "0xd594429c", "s16_stereo_22_5kHz.flac": "0x1d7b2d6d"},
            "lengths": {"src": "src/song/decoder/symphonia.rs:384-402,700-733 (expected_output_len = ceil(ratio * len), equal to FFmpeg's)",
                        "flush_test_52000.wav": 23888},
            "cue": {"src": "src/cue.rs:270-415 (assert_eq on the whole Song), indices from data/testcue.cue: 0:00:00, 0:11:05, 0:16:69",
                    "file": "testcue.flac", "index_mm_ss_ff": [[0, 0, 0], [0, 11, 5], [0, 16, 69]],
                    "tracks": [
                        [0.38463724, -0.85219246, -0.761946, -0.8904667, -0.63892543, -0.73945934, -0.80040205,
                         -0.82372904, 0.33865356, 0.32481194, -0.3433048, -0.6278722, -0.2809375, 0.08685577,
                         0.24455929, -0.5721703, 0.23292911, 0.19979906, -0.5859135, -0.06785172, -0.05990714,
                         -0.58482605, -0.078823924],
                        [0.18622077, -0.5989029, -0.5554645, -0.63438654, -0.24163479, -0.25766593, -0.40616918,
                         -0.23334831, 0.76875293, 0.7785741, -0.10609609, -0.14194643, -0.21418405, -0.21676934,
                         -0.20846015, -0.22077763, -0.0002696514, -0.00034928322, 0.0003143549, 0.00030446053,
                         -0.47109652, -0.66400576, 0.15099311],
                        [0.0024260283, 0.9874661, 0.97330654, -0.97244257, 0.99678576, -0.9961549, -0.98401415,
                         -0.9269961, 0.7498772, 0.22429907, 0.9990841, -0.9723601, -0.973079, -0.97307926,
                         -0.97308147, -0.9730794, -2.783537e-5, -2.7775764e-5, 3.1113625e-5, 2.4557114e-5,
                         -0.9210111, -0.99999785, -0.99993163]]},
            "cue_v1": {"src": "src/cue.rs:417-523 (test_cue_analysis_with_options: FeaturesVersion::Version1 on the same tracks)",
                       "tracks": [
                           [0.38463724, -0.85219246, -0.761946, -0.8904667, -0.63892543, -0.73945934, -0.80040205,
                            -0.82372904, 0.33865356, 0.32481194, -0.35692245, -0.6355889, -0.29584837, 0.06431806,
                            0.21875131, -0.58104205, -0.9466792, -0.94811195, -0.9820919, -0.9596871],
                           [0.18622077, -0.5989029, -0.5554645, -0.63438654, -0.24163479, -0.25766593, -0.40616918,
                            -0.23334831, 0.76875293, 0.7785741, -0.5075115, -0.5272629, -0.56706166, -0.568486,
                            -0.5639081, -0.5706943, -0.96501005, -0.96501285, -0.9649896, -0.96498996],
                           [0.0024260283, 0.9874661, 0.97330654, -0.97244257, 0.99678576, -0.9961549, -0.98401415,
                            -0.9269961, 0.7498772, 0.22429907, -0.8355152, -0.9977258, -0.9977849, -0.997785,
                            -0.99778515, -0.997785, -0.99999976, -0.99999976, -0.99999976, -0.99999976]]},
            "tone_11080Hz": {"_note": "SpectralDesc over chunks_exact(HOP_SIZE) of the decoded (resampled) file",
                             "centroid": {"src": "src/timbral.rs:430-439", "tol": 1e-5, "values": [0.97266, -0.9609926]},
                             "rolloff": {"src": "src/timbral.rs:364-373", "tol": 1e-4, "values": [0.9967681, -0.99615175]}},
            "analysis_symphonia_s32_stereo_44_1_kHz": {
                "src": "src/song/mod.rs:645-684 (the reference's OTHER decoder, rubato resampler: its own tolerance is 0.1)", "tol": 0.1,
                "values": [0.38463664, -0.85172224, -0.7607465, -0.8857495, -0.63906085, -0.73908424, -0.7890965,
                           -0.8191868, 0.33856833, 0.3246863, -0.34292227, -0.62803173, -0.2809453, 0.08687115,
                           0.2444489, -0.5723239, 0.23292565, 0.19979525, -0.58593845, -0.06783122, -0.060014784,
                           -0.5848569, -0.07879859]}},
    }
    with open(os.path.join(HERE, "reference_literals.json"), "w") as f:
        json.dump(literals, f, indent=1)
    print("fixtures written to", HERE)


if __name__ == "__main__":
    main()
