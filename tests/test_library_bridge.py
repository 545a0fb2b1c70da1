"""Feature store <-> dense matrix bridge: host code over SQLite, no GPU."""
import sqlite3

import numpy as np
import pytest


@pytest.fixture()
def db(tmp_path):
    import bliss_rs_amd as bliss

    path = str(tmp_path / "songs.db")
    bliss.library.create_schema(path)
    return path


def _song(bliss, i, version, rng, **kw):
    d = bliss.FeaturesVersion(version).feature_count()
    return bliss.Song(path=f"/music/{version}/{i:03d}.flac", title=f"t{i}", artist="a", album=f"al{i % 3}", track_number=i,
                      disc_number=1, duration=180.5 + i,
                      analysis=bliss.Analysis(rng.uniform(-1, 1, d).astype(np.float32), bliss.FeaturesVersion(version)),
                      features_version=bliss.FeaturesVersion(version), **kw)


def test_round_trip_is_bit_exact_and_filters_like_songs_from_library(db):
    import bliss_rs_amd as bliss

    rng = np.random.default_rng(1)
    v2 = [_song(bliss, i, 2, rng) for i in range(7)]
    v1 = [_song(bliss, i, 1, rng) for i in range(3)]
    bliss.library.store_songs(db, v2[:4] + v1 + v2[4:])          # interleaved ids
    conn = sqlite3.connect(db)
    conn.execute("insert into song (path, version, analyzed, error) values ('/music/broken.flac', 2, false, 'boom')")
    conn.commit()
    conn.close()

    ids, paths, m = bliss.library.load_feature_matrix(db, bliss.FeaturesVersion.Version2)
    assert m.dtype == np.float32 and m.shape == (7, 23)
    assert paths == [s.path for s in v2] and list(ids) == sorted(ids)        # order by song id, unanalysed song skipped
    assert np.array_equal(m.view(np.uint32), np.stack([s.analysis.as_arr1() for s in v2]).view(np.uint32))
    ids1, paths1, m1 = bliss.library.load_feature_matrix(db, bliss.FeaturesVersion.Version1)
    assert m1.shape == (3, 20) and paths1 == [s.path for s in v1]

    songs = bliss.library.load_songs(db, 2)
    assert [s.path for s in songs] == paths
    assert songs[2].analysis == v2[2].analysis and songs[2].album == v2[2].album and songs[2].track_number == 2
    assert songs[2].duration == v2[2].duration

    # storing a song again replaces its features (src/library.rs:1612-1627)
    again = _song(bliss, 2, 2, rng)
    bliss.library.store_song(db, again)
    _, _, m2 = bliss.library.load_feature_matrix(db, 2)
    assert np.array_equal(m2[2], again.analysis.as_arr1()) and np.array_equal(np.delete(m2, 2, 0), np.delete(m, 2, 0))


def test_wrong_feature_count_is_a_provider_error(db):
    import bliss_rs_amd as bliss

    rng = np.random.default_rng(2)
    bliss.library.store_song(db, _song(bliss, 0, 2, rng))
    conn = sqlite3.connect(db)
    conn.execute("delete from feature where feature_index = 22")
    conn.commit()
    conn.close()
    with pytest.raises(bliss.ProviderError, match="does not match the expected version feature count 23"):
        bliss.library.load_feature_matrix(db, 2)
    ids, paths, m = bliss.library.load_feature_matrix(db, 1)   # nothing stored for version 1: empty, not an error
    assert m.shape == (0, 20) and paths == []
