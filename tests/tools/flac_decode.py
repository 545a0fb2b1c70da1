"""Minimal pure-Python FLAC decoder -- TEST TOOLING ONLY (never on the product path).

Used once, in this container, by tests/golden/make_fixtures.py to recover the exact PCM
the reference's golden feature vectors were computed from (the reference pins that PCM by
Adler-32 over the f32le stream: src/song/decoder/ffmpeg.rs:455-462 -> 0x5e01930b for
data/s16_mono_22_5kHz.flac, :524-527 -> 0xde831e82 for data/piano.flac).

Supports what those files need: STREAMINFO, fixed/variable block headers, subframes
CONSTANT / VERBATIM / FIXED(0..4) / LPC(1..32), wasted bits, Rice residuals (4- and 5-bit
parameters, escape partitions), independent / left-side / side-right / mid-side channels.
CRCs are skipped (acceptance is the Adler-32 check in make_fixtures.py).
"""
import zlib

import numpy as np


class _Bits:
    __slots__ = ("d", "p", "acc", "n")

    def __init__(self, data, pos):
        self.d = data
        self.p = pos  # next byte to load
        self.acc = 0
        self.n = 0  # valid bits in acc

    def read(self, k):
        if k == 0:
            return 0
        while self.n < k:
            self.acc = (self.acc << 8) | self.d[self.p]
            self.p += 1
            self.n += 8
        self.n -= k
        v = (self.acc >> self.n) & ((1 << k) - 1)
        self.acc &= (1 << self.n) - 1
        return v

    def read_signed(self, k):
        v = self.read(k)
        return v - (1 << k) if k and (v >> (k - 1)) else v

    def unary(self):
        # number of 0 bits before the next 1 bit
        q = 0
        while True:
            if self.n == 0:
                self.acc = self.d[self.p]
                self.p += 1
                self.n = 8
            if self.acc == 0:
                q += self.n
                self.n = 0
                continue
            lead = self.n - self.acc.bit_length()
            q += lead
            self.n -= lead + 1
            self.acc &= (1 << self.n) - 1
            return q

    def align(self):
        self.n = 0
        self.acc = 0

    def byte_pos(self):
        assert self.n % 8 == 0
        return self.p - self.n // 8


_FIXED = {
    0: (),
    1: (1,),
    2: (2, -1),
    3: (3, -3, 1),
    4: (4, -6, 4, -1),
}


def _residual(br, blocksize, order, out):
    method = br.read(2)
    assert method in (0, 1)
    pbits = 4 if method == 0 else 5
    esc = (1 << pbits) - 1
    porder = br.read(4)
    nparts = 1 << porder
    for part in range(nparts):
        cnt = (blocksize >> porder) - (order if part == 0 else 0)
        k = br.read(pbits)
        if k == esc:
            nb = br.read(5)
            for _ in range(cnt):
                out.append(br.read_signed(nb) if nb else 0)
        else:
            for _ in range(cnt):
                q = br.unary()
                v = (q << k) | br.read(k)
                out.append((v >> 1) ^ -(v & 1))


def _subframe(br, blocksize, bps):
    assert br.read(1) == 0
    typ = br.read(6)
    wasted = 0
    if br.read(1):
        wasted = br.unary() + 1
    bps -= wasted
    if typ == 0:
        s = [br.read_signed(bps)] * blocksize
    elif typ == 1:
        s = [br.read_signed(bps) for _ in range(blocksize)]
    elif 8 <= typ <= 12:
        order = typ - 8
        s = [br.read_signed(bps) for _ in range(order)]
        res = []
        _residual(br, blocksize, order, res)
        c = _FIXED[order]
        for r in res:
            pred = 0
            for j, cj in enumerate(c):
                pred += cj * s[-1 - j]
            s.append(pred + r)
    elif typ >= 32:
        order = (typ & 31) + 1
        s = [br.read_signed(bps) for _ in range(order)]
        prec = br.read(4) + 1
        shift = br.read_signed(5)
        coef = [br.read_signed(prec) for _ in range(order)]
        res = []
        _residual(br, blocksize, order, res)
        for r in res:
            pred = 0
            for j in range(order):
                pred += coef[j] * s[-1 - j]
            s.append((pred >> shift) + r)
    else:
        raise ValueError(f"reserved subframe type {typ}")
    if wasted:
        s = [x << wasted for x in s]
    return s


def decode_flac(path):
    """Return (samples int64 [n, channels], sample_rate, bits_per_sample)."""
    data = open(path, "rb").read()
    assert data[:4] == b"fLaC"
    pos = 4
    info = None
    while True:
        hdr = data[pos]
        length = int.from_bytes(data[pos + 1:pos + 4], "big")
        body = data[pos + 4:pos + 4 + length]
        if (hdr & 0x7F) == 0:
            v = int.from_bytes(body[10:18], "big")
            info = dict(
                sample_rate=v >> 44,
                channels=((v >> 41) & 7) + 1,
                bps=((v >> 36) & 31) + 1,
                total=v & ((1 << 36) - 1),
            )
        pos += 4 + length
        if hdr & 0x80:
            break
    ch, bps_stream, total = info["channels"], info["bps"], info["total"]
    out = [[] for _ in range(ch)]
    got = 0
    while got < total:
        br = _Bits(data, pos)
        sync = br.read(14)
        assert sync == 0x3FFE, (hex(sync), pos)
        br.read(1)
        br.read(1)  # blocking strategy
        bs_code = br.read(4)
        sr_code = br.read(4)
        ch_assign = br.read(4)
        ss_code = br.read(3)
        br.read(1)
        # UTF-8-like coded number
        first = br.read(8)
        nfollow = 0
        while first & (0x80 >> nfollow):
            nfollow += 1
        for _ in range(max(0, nfollow - 1)):
            br.read(8)
        if bs_code == 1:
            blocksize = 192
        elif 2 <= bs_code <= 5:
            blocksize = 576 << (bs_code - 2)
        elif bs_code == 6:
            blocksize = br.read(8) + 1
        elif bs_code == 7:
            blocksize = br.read(16) + 1
        else:
            blocksize = 256 << (bs_code - 8)
        if sr_code == 12:
            br.read(8)
        elif sr_code in (13, 14):
            br.read(16)
        br.read(8)  # CRC-8
        bps = {0: bps_stream, 1: 8, 2: 12, 4: 16, 5: 20, 6: 24, 7: 32}[ss_code]
        if ch_assign < 8:
            subs = [_subframe(br, blocksize, bps) for _ in range(ch_assign + 1)]
        elif ch_assign == 8:  # left, side
            l = _subframe(br, blocksize, bps)
            s = _subframe(br, blocksize, bps + 1)
            subs = [l, [a - b for a, b in zip(l, s)]]
        elif ch_assign == 9:  # side, right
            s = _subframe(br, blocksize, bps + 1)
            r = _subframe(br, blocksize, bps)
            subs = [[a + b for a, b in zip(s, r)], r]
        elif ch_assign == 10:  # mid, side
            m = _subframe(br, blocksize, bps)
            s = _subframe(br, blocksize, bps + 1)
            l, r = [], []
            for a, b in zip(m, s):
                a = (a << 1) | (b & 1)
                l.append((a + b) >> 1)
                r.append((a - b) >> 1)
            subs = [l, r]
        else:
            raise ValueError("reserved channel assignment")
        br.align()
        pos = br.byte_pos() + 2  # CRC-16
        for c in range(ch):
            out[c].extend(subs[c])
        got += blocksize
    arr = np.array(out, dtype=np.int64).T[:total]
    return arr, info["sample_rate"], bps_stream


def adler32_f32le(x):
    return zlib.adler32(np.asarray(x, dtype="<f4").tobytes()) & 0xFFFFFFFF


if __name__ == "__main__":
    import sys

    a, sr, bps = decode_flac(sys.argv[1])
    pcm = (a[:, 0] / float(1 << (bps - 1))).astype(np.float32)
    print(a.shape, sr, bps, hex(adler32_f32le(pcm)))
