"""The animation driver's last step (trace_of_radiance_animation.nim:203-210, io/mp4.nim:113-163): the Annex-B
H.264 stream wrapped into an MP4 file.  No player or demuxer is installed here, so the test parses the file per
ISO/IEC 14496-12/-15 itself: box tree, avcC = the stream's SPS/PPS, one sample per slice NAL unit found through
stsz/stsc/co64 (4-byte length + the NAL bytes), 90 kHz time base with 90000/fps ticks per frame.  CPU only."""
import struct

import numpy as np
import pytest


def _boxes(buf, start=0, end=None):
    end = len(buf) if end is None else end
    out = []
    pos = start
    while pos < end:
        size, tag = struct.unpack(">I4s", buf[pos:pos + 8])
        hdr = 8
        if size == 1:
            size = struct.unpack(">Q", buf[pos + 8:pos + 16])[0]
            hdr = 16
        assert size >= hdr and pos + size <= end, (tag, size)
        out.append((tag.decode("latin1"), pos + hdr, pos + size))
        pos += size
    assert pos == end
    return out


def _child(buf, boxes, tag):
    hits = [b for b in boxes if b[0] == tag]
    assert len(hits) == 1, (tag, [b[0] for b in boxes])
    return hits[0]


def _nals(stream):
    parts, i, n = [], 0, len(stream)
    starts = []
    while i + 3 <= n:
        if stream[i:i + 3] == b"\x00\x00\x01":
            starts.append(i)
            i += 3
        else:
            i += 1
    for k, s in enumerate(starts):
        e = starts[k + 1] if k + 1 < len(starts) else n
        parts.append(stream[s + 3:e].rstrip(b"\x00"))
    return parts


@pytest.mark.parametrize("w,h,frames,fps", [(32, 16, 3, 30), (64, 48, 5, 25)])
def test_mp4_wraps_the_stream(tor, oracle, tmp_path, w, h, frames, fps):
    rng = np.random.default_rng(w)
    stream = oracle.h264_stream_header(w, h)
    for _ in range(frames):
        stream += oracle.encode_frame(rng.uniform(0.0, 1.1, (h, w, 3)))[4]
    src, dst = tmp_path / "a.264", tmp_path / "a.mp4"
    src.write_bytes(stream)
    assert tor.mp4_mux_file(str(src), str(dst), w, h, fps) == frames
    mp4 = dst.read_bytes()
    nals = _nals(stream)
    sps = [x for x in nals if x[0] & 0x1f == 7][0]
    pps = [x for x in nals if x[0] & 0x1f == 8][0]
    slices = [x for x in nals if x[0] & 0x1f in (1, 5)]
    assert len(slices) == frames

    top = _boxes(mp4)
    assert [t for t, _, _ in top] == ["ftyp", "mdat", "moov"]
    _, f0, f1 = top[0]
    assert mp4[f0:f0 + 4] == b"isom" and b"avc1" in mp4[f0 + 8:f1]
    _, md0, md1 = top[1]
    moov = _boxes(mp4, *top[2][1:])
    mvhd = _child(mp4, moov, "mvhd")
    assert mp4[mvhd[1]] == 1                                                     # version 1: 64-bit times
    timescale, duration = struct.unpack(">IQ", mp4[mvhd[1] + 20:mvhd[1] + 32])
    assert timescale == 90000 and duration == frames * (90000 // fps)
    trak = _boxes(mp4, *_child(mp4, moov, "trak")[1:])
    tkhd = _child(mp4, trak, "tkhd")
    assert struct.unpack(">II", mp4[tkhd[2] - 8:tkhd[2]]) == (w << 16, h << 16)
    mdia = _boxes(mp4, *_child(mp4, trak, "mdia")[1:])
    mdhd = _child(mp4, mdia, "mdhd")
    assert struct.unpack(">IQ", mp4[mdhd[1] + 20:mdhd[1] + 32]) == (90000, frames * (90000 // fps))
    hdlr = _child(mp4, mdia, "hdlr")
    assert mp4[hdlr[1] + 8:hdlr[1] + 12] == b"vide"
    minf = _boxes(mp4, *_child(mp4, mdia, "minf")[1:])
    assert {t for t, _, _ in minf} == {"vmhd", "dinf", "stbl"}
    stbl = _boxes(mp4, *_child(mp4, minf, "stbl")[1:])
    names = [t for t, _, _ in stbl]
    assert names == ["stsd", "stts", "stsc", "stsz", "co64"]                     # every sample is an IDR: no stss

    stsd = _child(mp4, stbl, "stsd")
    assert struct.unpack(">I", mp4[stsd[1] + 4:stsd[1] + 8])[0] == 1
    entry = _boxes(mp4, stsd[1] + 8, stsd[2])
    assert entry[0][0] == "avc1"
    a0 = entry[0][1]
    assert struct.unpack(">HH", mp4[a0 + 24:a0 + 28]) == (w, h)
    avcc = _boxes(mp4, a0 + 78, entry[0][2])
    assert avcc[0][0] == "avcC"
    c = mp4[avcc[0][1]:avcc[0][2]]
    assert c[0] == 1 and c[1:4] == sps[1:4] and c[4] == 0xff and c[5] == 0xe1
    n_sps = struct.unpack(">H", c[6:8])[0]
    assert c[8:8 + n_sps] == sps
    q = 8 + n_sps
    assert c[q] == 1
    n_pps = struct.unpack(">H", c[q + 1:q + 3])[0]
    assert c[q + 3:q + 3 + n_pps] == pps and q + 3 + n_pps == len(c)

    stts = _child(mp4, stbl, "stts")
    assert struct.unpack(">III", mp4[stts[1] + 4:stts[1] + 16]) == (1, frames, 90000 // fps)
    stsc = _child(mp4, stbl, "stsc")
    assert struct.unpack(">IIII", mp4[stsc[1] + 4:stsc[1] + 20]) == (1, 1, 1, 1)
    stsz = _child(mp4, stbl, "stsz")
    const, count = struct.unpack(">II", mp4[stsz[1] + 4:stsz[1] + 12])
    assert count == frames
    sizes = [const] * frames if const else list(struct.unpack(f">{frames}I", mp4[stsz[1] + 12:stsz[1] + 12 + 4 * frames]))
    co64 = _child(mp4, stbl, "co64")
    assert struct.unpack(">I", mp4[co64[1] + 4:co64[1] + 8])[0] == frames
    offs = struct.unpack(f">{frames}Q", mp4[co64[1] + 8:co64[1] + 8 + 8 * frames])
    for k in range(frames):
        assert md0 <= offs[k] and offs[k] + sizes[k] <= md1
        assert struct.unpack(">I", mp4[offs[k]:offs[k] + 4])[0] == len(slices[k]) == sizes[k] - 4
        assert mp4[offs[k] + 4:offs[k] + sizes[k]] == slices[k]
    assert offs[0] == md0 and offs[-1] + sizes[-1] == md1                        # mdat holds the samples and nothing else


def test_mp4_rejects_bad_input(tor, oracle, tmp_path):
    dst = tmp_path / "x.mp4"
    with pytest.raises(tor.TorError):
        tor.mp4_mux_file(str(tmp_path / "missing.264"), str(dst), 32, 16)
    empty = tmp_path / "empty.264"
    empty.write_bytes(b"")
    with pytest.raises(tor.TorError):
        tor.mp4_mux_file(str(empty), str(dst), 32, 16)
    assert not dst.exists()
    hdr_only = tmp_path / "hdr.264"
    hdr_only.write_bytes(oracle.h264_stream_header(32, 16))
    with pytest.raises(tor.TorError):
        tor.mp4_mux_file(str(hdr_only), str(dst), 32, 16)
    with pytest.raises(tor.TorError):
        tor.mp4_mux_file(str(hdr_only), str(dst), 32, 16, fps=0)
    assert not dst.exists()


def test_mp4_large_units_cross_the_read_window(tor, oracle, tmp_path):
    """Frames larger than the reader's 1 MiB window (the 1080p case is 3 MB per frame)."""
    w, h = 1280, 720
    stream = oracle.h264_stream_header(w, h)
    frame = oracle.encode_frame(np.full((h, w, 3), 0.5))[4]
    assert len(frame) > (1 << 20)
    stream += frame + frame
    src, dst = tmp_path / "big.264", tmp_path / "big.mp4"
    src.write_bytes(stream)
    assert tor.mp4_mux_file(str(src), str(dst), w, h, 30) == 2
    mp4 = dst.read_bytes()
    body = frame[3:] if frame[:3] == b"\x00\x00\x01" else frame[4:]
    assert mp4.count(struct.pack(">I", len(body)) + body[:64]) == 2
