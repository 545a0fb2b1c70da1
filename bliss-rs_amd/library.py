"""Feature store <-> dense feature matrix.

An existing bliss-rs SQLite library (`Library`, src/library.rs) keeps one row per (song, feature index) in the table
`feature(song_id, feature real, feature_index)` next to `song(id, path, ..., version, analyzed, ...)`
(src/library.rs:500-531).  The playlist / pairwise kernels want an n x d f32 matrix; these helpers move between the
two without re-analysing anything:

    load_feature_matrix   the read path of `songs_from_library` (src/library.rs:1355-1372): songs with
                          analyzed = true and version = ?, ordered by song id, features ordered by feature_index
    load_songs            the same rows as `Song` objects (metadata + Analysis)
    store_song            the write path of `store_song` (src/library.rs:1560-1630): upsert the song row, replace
                          its feature rows
    create_schema         the two tables, for new databases (user_version = the number of migrations)
    upgrade               `Library::upgrade` (src/library.rs:631-681): brings a database written by an older bliss-rs to
                          the current schema (track_number text -> integer, disc_number, training_triplet,
                          version not null default 1), keyed on `pragma user_version` like the crate

SQLite stores `real` as f64; an f32 feature widens exactly on the way in and narrows exactly on the way out, so a
round trip is bit-exact.  Pure host code: nothing here touches the GPU.
"""
import sqlite3
from typing import List, Sequence, Tuple, Union

import numpy as np

from .song import Analysis, FeaturesVersion, ProviderError, Song

_SONG_COLUMNS = ("path", "artist", "title", "album", "album_artist", "track_number", "disc_number", "genre", "duration",
                 "version")

Conn = Union[str, sqlite3.Connection]


def _connect(db: Conn) -> Tuple[sqlite3.Connection, bool]:
    if isinstance(db, sqlite3.Connection):
        return db, False
    return sqlite3.connect(db), True


def create_schema(db: Conn) -> None:
    """Tables `song` and `feature` with the columns of src/library.rs:500-531."""
    conn, own = _connect(db)
    try:
        conn.executescript(
            """
            create table if not exists song (
                id integer primary key, path text not null unique, duration float, album_artist text, artist text,
                title text, album text, track_number integer, disc_number integer, genre text, cue_path text,
                audio_file_path text, stamp timestamp default current_timestamp, version integer not null,
                analyzed boolean default false, extra_info json, error text);
            pragma foreign_keys = on;
            create table if not exists feature (
                id integer primary key, song_id integer not null, feature real not null, feature_index integer not null,
                unique(song_id, feature_index), foreign key(song_id) references song(id) on delete cascade);
            """)
        conn.execute("pragma user_version = 5")  # = len(SQLITE_MIGRATIONS): a new database needs no migration
        conn.commit()
    finally:
        if own:
            conn.close()


# SQLITE_MIGRATIONS of the crate (src/library.rs:532-606), by schema version; the statements are the schema contract an
# existing library file obeys, so they are restated as they are.
_MIGRATIONS = (
    "",
    """
    alter table song add column track_number_1 integer;
    update song set track_number_1 = s1.cast_track_number from (
        select cast(track_number as int) as cast_track_number, id from song
    ) as s1 where s1.id = song.id and cast(track_number as int) != 0;
    alter table song drop column track_number;
    alter table song rename column track_number_1 to track_number;
    """,
    "alter table song add column disc_number integer;",
    """
    create table training_triplet (
        id integer primary key, song_1_id integer not null, song_2_id integer not null,
        odd_one_out_id integer not null, stamp timestamp default current_timestamp,
        foreign key(song_1_id) references song(id) on delete cascade,
        foreign key(song_2_id) references song(id) on delete cascade,
        foreign key(odd_one_out_id) references song(id) on delete cascade)
    """,
    """
    create table song_bak (
        id integer primary key, path text not null unique, duration float, album_artist text, artist text,
        title text, album text, track_number integer, disc_number integer, genre text, cue_path text,
        audio_file_path text, stamp timestamp default current_timestamp, version integer not null,
        analyzed boolean default false, extra_info json, error text);
    insert into song_bak (id, path, duration, album_artist, artist, title, album, track_number, disc_number, genre,
                          cue_path, audio_file_path, stamp, version, analyzed, extra_info, error)
        select id, path, duration, album_artist, artist, title, album, track_number, disc_number, genre, cue_path,
               audio_file_path, stamp, coalesce(version, 1), analyzed, extra_info, error from song;
    drop table song;
    alter table song_bak rename to song;
    """,
)


def upgrade(db: Conn) -> int:
    """`Library::upgrade` (src/library.rs:631-681): run the migrations an older database is missing; returns the schema
    version it ends at (5).  A database newer than this mirror is a ProviderError, an empty one gets the current schema."""
    conn, own = _connect(db)
    try:
        version = conn.execute("pragma user_version").fetchone()[0]
        if version > len(_MIGRATIONS):
            raise ProviderError(f"bliss-rs version {version} is older than the schema version {len(_MIGRATIONS)}")
        if version == len(_MIGRATIONS):
            return version
        tables = conn.execute("select count(*) from sqlite_master where type = 'table'").fetchone()[0]
        if version == 0 and tables == 0:
            create_schema(conn)
        else:
            for migration in _MIGRATIONS[version:]:
                conn.executescript(migration)
        conn.execute(f"pragma user_version = {len(_MIGRATIONS)}")
        conn.commit()
        return len(_MIGRATIONS)
    finally:
        if own:
            conn.close()


def load_feature_matrix(db: Conn, features_version: FeaturesVersion = FeaturesVersion.LATEST):
    """-> (song_ids int64[n], paths list[str], matrix float32[n, d]) for the analysed songs of that features version."""
    version = FeaturesVersion(features_version)
    d = version.feature_count()
    conn, own = _connect(db)
    try:
        songs = conn.execute("select id, path from song where analyzed = true and version = ? order by id",
                             (int(version),)).fetchall()
        rows = conn.execute(
            "select feature, song.id from feature join song on song.id = feature.song_id "
            "where song.analyzed = true and song.version = ? order by song_id, feature_index", (int(version),)).fetchall()
    finally:
        if own:
            conn.close()
    ids = np.array([s[0] for s in songs], np.int64)
    feats = np.array([r[0] for r in rows], np.float64)
    owner = np.array([r[1] for r in rows], np.int64)
    # _songs_from_statement (src/library.rs:1297-1345) groups the feature rows by song id and Analysis::new rejects a
    # song that does not carry exactly feature_count() of them, naming the first offender
    counts = {int(i): 0 for i in ids}
    for o in owner:
        counts[int(o)] = counts.get(int(o), 0) + 1
    for (song_id, path) in songs:
        if counts.get(int(song_id), 0) != d:
            raise ProviderError(f"Song with ID {song_id} and path {path} has a different feature number than expected. "
                                "Please rescan or update the song library.")
    return ids, [s[1] for s in songs], feats.astype(np.float32).reshape(ids.size, d)


def load_songs(db: Conn, features_version: FeaturesVersion = FeaturesVersion.LATEST) -> List[Song]:
    """`songs_from_library` (src/library.rs:1355-1372) without the extra_info payload."""
    version = FeaturesVersion(features_version)
    conn, own = _connect(db)
    try:
        ids, _, matrix = load_feature_matrix(conn, version)
        meta = conn.execute(
            f"select {', '.join(_SONG_COLUMNS)}, id from song where analyzed = true and version = ? order by id",
            (int(version),)).fetchall()
    finally:
        if own:
            conn.close()
    out = []
    for row, feats in zip(meta, matrix):
        kw = dict(zip(_SONG_COLUMNS, row[:-1]))
        kw.pop("version")
        kw["duration"] = float(kw["duration"] or 0.0)
        out.append(Song(analysis=Analysis(feats, version), features_version=version, **kw))
    return out


def store_song(db: Conn, song: Song) -> None:
    """`Library::store_song` (src/library.rs:1560-1630): upsert the song (analyzed = true), replace its features."""
    conn, own = _connect(db)
    try:
        version = FeaturesVersion(song.features_version)
        conn.execute(
            "insert into song (path, artist, title, album, album_artist, track_number, disc_number, genre, duration, "
            "analyzed, version) values (?, ?, ?, ?, ?, ?, ?, ?, ?, true, ?) "
            "on conflict(path) do update set artist=excluded.artist, title=excluded.title, album=excluded.album, "
            "album_artist=excluded.album_artist, track_number=excluded.track_number, disc_number=excluded.disc_number, "
            "genre=excluded.genre, duration=excluded.duration, analyzed=excluded.analyzed, version=excluded.version",
            (song.path, song.artist, song.title, song.album, song.album_artist, song.track_number, song.disc_number,
             song.genre, float(song.duration), int(version)))
        conn.execute("delete from feature where song_id in (select id from song where path = ?)", (song.path,))
        conn.executemany(
            "insert into feature (song_id, feature, feature_index) values ((select id from song where path = ?), ?, ?) "
            "on conflict(song_id, feature_index) do update set feature=excluded.feature",
            [(song.path, float(np.float32(v)), i) for i, v in enumerate(song.analysis.as_vec())])
        conn.commit()
    finally:
        if own:
            conn.close()


def store_songs(db: Conn, songs: Sequence[Song]) -> None:
    conn, own = _connect(db)
    try:
        for s in songs:
            store_song(conn, s)
    finally:
        if own:
            conn.close()
