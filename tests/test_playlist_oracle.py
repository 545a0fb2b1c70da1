"""Playlist ordering: the oracle's restatement of src/playlist.rs:24-59, 173-221, 256-326,
367-402 pinned on the reference's own ordering tests (tests/golden/playlist_cases.json holds their inputs and
expected outputs), plus the host-side mirror pieces that need no GPU."""
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = json.load(open(os.path.join(ROOT, "tests", "golden", "playlist_cases.json")))
SONGS = CASES["songs"]


def mat(names):
    return np.array([SONGS[n]["analysis"] for n in names], np.float32)


def same_meta(names):
    n = len(names)
    sm = np.zeros((n, n), np.uint8)
    for i, a in enumerate(names):
        for j, b in enumerate(names):
            A, B = SONGS[a], SONGS[b]
            sm[i, j] = int(all(k in A and k in B for k in ("title", "artist")) and A["title"] == B["title"]
                           and A["artist"] == B["artist"])
    return sm


@pytest.mark.parametrize("case", CASES["closest_to_songs"])
def test_oracle_closest_to_songs_reference_cases(oracle, case):  # src/playlist.rs:860-1007
    order, dist = oracle.closest_to_songs(mat(case["initial"]), mat(case["candidates"]), case["metric"])
    assert [case["candidates"][i] for i in order] == case["expected"]
    assert np.all(np.diff(dist[order]) >= 0)


@pytest.mark.parametrize("case", CASES["song_to_song"])
def test_oracle_song_to_song_reference_cases(oracle, case):  # src/playlist.rs:733-858
    order = oracle.song_to_song(mat(case["initial"]), mat(case["candidates"]), case["metric"])
    assert [case["candidates"][i] for i in order] == case["expected"]


@pytest.mark.parametrize("case", CASES["dedup"])
def test_oracle_dedup_reference_cases(oracle, case):  # src/playlist.rs:506-731
    pl = case["playlist"]
    thr = 0.05 if case["threshold"] is None else case["threshold"]
    kept = oracle.dedup_playlist(mat(pl), thr, case["metric"], same_meta=same_meta(pl))
    assert [pl[i] for i in kept] == case["expected"]


def test_oracle_variance_weight_matrix_reference_asserts(oracle):  # src/playlist.rs:1663-1765
    v = CASES["variance"]
    m = oracle.variance_based_weight_matrix(np.array(v["stable_vs_variable"], np.float32))
    assert m.shape == (3, 3)
    assert m[0, 0] > m[1, 1] and m[2, 2] > m[1, 1]
    assert np.count_nonzero(m - np.diag(np.diag(m))) == 0
    assert abs(float(np.trace(m)) - 3.0) < 1e-4
    m = oracle.variance_based_weight_matrix(np.array(v["identical"], np.float32))
    assert np.all(np.abs(np.diag(m) - 1.0) < 1e-4)
    m = oracle.variance_based_weight_matrix(np.array(v["two_seeds"], np.float32))
    assert m.shape == (2, 2) and m[0, 0] > m[1, 1]
    with pytest.raises(ValueError, match="more than one element"):
        oracle.variance_based_weight_matrix(np.array([[1.0, 2.0, 3.0]], np.float32))


def test_host_variance_weight_matrix_matches_oracle_bit_for_bit(oracle):
    import bliss_rs_amd as bliss

    rng = np.random.default_rng(5)
    for n_seeds, d in ((2, 2), (3, 3), (5, 23), (17, 20), (4, 40)):
        seeds = rng.standard_normal((n_seeds, d)).astype(np.float32) * rng.uniform(0.01, 50, d).astype(np.float32)
        got = bliss.playlist.variance_based_weight_matrix(list(seeds))
        ref = oracle.variance_based_weight_matrix(seeds)
        assert got.dtype == np.float32 and np.array_equal(got, ref)


def test_host_variance_weight_matrix_errors():  # the three ProviderError messages, src/playlist.rs:174-190
    import bliss_rs_amd as bliss

    with pytest.raises(bliss.ProviderError, match="seeds must contain more than one element"):
        bliss.playlist.variance_based_weight_matrix([[1.0, 2.0, 3.0]])
    with pytest.raises(bliss.ProviderError, match="all seed feature vectors must have the same length"):
        bliss.playlist.variance_based_weight_matrix([[1.0, 2.0, 3.0], [1.0, 2.0]])
    with pytest.raises(bliss.ProviderError, match="seed feature vectors must not be empty"):
        bliss.playlist.variance_based_weight_matrix([[], []])


def test_metric_builder_rejects_arbitrary_callables():
    import bliss_rs_amd as bliss

    with pytest.raises(TypeError, match="evaluated on the GPU"):
        bliss.playlist.closest_to_songs([], [object()], lambda a, b: 0.0)


def test_oracle_orders_are_consistent_on_random_data(oracle):
    """closest_to_songs is a stable sort of set distances; song_to_song's first element is their first argmin."""
    rng = np.random.default_rng(11)
    X = rng.uniform(-1, 1, (300, 23)).astype(np.float32)
    X[50] = X[10]  # exact ties
    X[200] = X[10]
    seeds = X[[10, 20, 30]]
    M = oracle.feature_weights(2)
    for metric, m in (("euclidean", None), ("cosine", None), ("mahalanobis", M)):
        order, dist = oracle.closest_to_songs(seeds, X, metric, m)
        assert sorted(order.tolist()) == list(range(300))
        expect = np.array([oracle.set_distance(seeds, x, metric, m) for x in X], np.float32)
        assert np.array_equal(dist, expect)
        assert np.array_equal(order, np.argsort(expect, kind="stable").astype(np.uint32))
        chain = oracle.song_to_song(seeds, X, metric, m)
        assert sorted(chain.tolist()) == list(range(300))
        assert chain[0] == int(np.argmin(expect))
