"""SURVEY 8 f2/f3 cannot be pinned to the reference (it holds no encoded frame, and its colour self-test needs an
absent yuv_rgb.c), so these tests check the video output stage against things the builder's encoder did NOT define:
  * tests/h264_spec_decoder.py -- an H.264 decoder for baseline / CAVLC / I_PCM streams written from the syntax tables
    of ITU-T H.264 (Annex B, 7.3.2.1.1, 7.3.2.2, 7.3.3, 7.3.4, 7.3.5): SPS, PPS, slice header, macroblock layer and
    trailing bits must parse the way a conforming decoder parses them, and the decoded planes must be the planes;
  * ITU-R BT.601 in floating point (limited range, 4:2:0 by 2x2 averaging): the reference's fixed-point conversion
    (io/color_conversions.nim:180-252) truncates three times (>> 8, >> 7, >> 2): it must stay within 2.1 code values for luma and 3 for chroma (measured and
    printed);
  * ISO/IEC 14496-12/-15: the MP4's sample table must lead to length-prefixed NAL units that, put back behind start
    codes together with the avcC parameter sets, decode to the same planes.
(No PyAV / ffprobe / ffmpeg in this image and no network to fetch them.)"""
import struct

import numpy as np
import pytest

from h264_spec_decoder import bt601_limited_float, decode_stream, split_annexb
from test_mp4 import _boxes, _child


def _check_parameter_sets(sps, pps, w, h):
    assert sps["profile_idc"] == 66 and sps["level_idc"] == 10                      # h264.nim:100-104
    assert (sps["pic_width_in_mbs"], sps["pic_height_in_map_units"]) == ((w + 15) // 16, (h + 15) // 16)
    assert sps["frame_mbs_only"] == 1 and sps["poc_type"] == 0 and sps["max_num_ref_frames"] == 0
    assert not sps["frame_cropping"] and not sps["vui_present"]
    assert pps["entropy_coding_mode"] == 0 and pps["num_slice_groups"] == 1 and pps["pps_id"] == 0 and pps["sps_id"] == sps["sps_id"]


def _colour_errors(rgb, Y, Cb, Cr):
    fy, fcb, fcr = bt601_limited_float(rgb)
    return (float(np.abs(Y.astype(np.float64) - fy).max()), float(np.abs(Cb.astype(np.float64) - fcb).max()),
            float(np.abs(Cr.astype(np.float64) - fcr).max()))


def test_stream_decodes_with_the_spec_decoder(tor, oracle):
    rng = np.random.default_rng(5)
    for w, h in ((16, 16), (48, 32), (256, 144)):
        frames = [rng.uniform(-0.1, 1.1, (h, w, 3)) for _ in range(3)]
        frames[0][:] = rng.uniform(0, 1, 3)                       # a flat frame
        enc = [oracle.encode_frame(f) for f in frames]
        for hdr in (oracle.h264_stream_header(w, h), tor.h264_stream_header(w, h)):
            stream = hdr + b"".join(e[4] for e in enc)
            sps, pps, pics = decode_stream(stream)
            _check_parameter_sets(sps, pps, w, h)
            assert len(pics) == len(frames)
            for (hd, Y, Cb, Cr), (rgb, oY, oCb, oCr, _) in zip(pics, enc):
                assert np.array_equal(Y, oY) and np.array_equal(Cb, oCb) and np.array_equal(Cr, oCr)
                assert hd["slice_type"] == 7 and hd["idr_pic_id"] == 0 and hd["slice_qp_delta"] == 0
                # the ONLY departure from the standard the decoder finds is the reference's own (h264.nim:38: nal_ref_idc 0
                # on an IDR slice); everything else parses as a conforming baseline stream
                assert hd["quirks"] == ["idr_with_nal_ref_idc_0"]
    # a decoder must also REJECT what is not a conforming stream: a flipped mb_type, a missing stop bit
    good = oracle.h264_stream_header(16, 16) + oracle.encode_frame(np.full((16, 16, 3), 0.5))[4]
    decode_stream(good)
    bad = bytearray(good); bad[-1] = 0x00
    with pytest.raises(AssertionError):
        decode_stream(bytes(bad))
    bad = bytearray(good); bad[len(oracle.h264_stream_header(16, 16)) + 8] ^= 0x40   # inside the first mb_type / alignment bits
    with pytest.raises(AssertionError):
        decode_stream(bytes(bad))


def test_fixed_point_bt601_tracks_the_float_matrix(oracle):
    """io/color_conversions.nim:180-252 (kr, kg, kb = 77, 150, 29; y_scale 110; fb, fr = 127, 160) against BT.601 in
    float64: primaries, greys, and random images."""
    rng = np.random.default_rng(11)
    worst = [0.0, 0.0, 0.0]
    imgs = [rng.uniform(0, 1, (64, 64, 3)) for _ in range(6)]
    for c in ((1, 1, 1), (0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 1, 0), (0, 1, 1), (1, 0, 1), (.5, .5, .5), (.2, .7, .4)):
        px = np.empty((32, 32, 3)); px[:] = c
        imgs.append(px)
    ramp = np.zeros((256, 16, 3)); ramp[..., :] = (np.arange(256)[:, None, None] + 0.5) / 256.0   # all 256 grey levels
    imgs.append(ramp)
    for px in imgs:
        rgb, Y, Cb, Cr, _ = oracle.encode_frame(px)
        e = _colour_errors(rgb, Y, Cb, Cr)
        worst = [max(a, b) for a, b in zip(worst, e)]
    print(f"fixed-point vs float BT.601: max |dY| = {worst[0]:.3f}, |dCb| = {worst[1]:.3f}, |dCr| = {worst[2]:.3f} code values")
    assert worst[0] <= 2.1 and worst[1] <= 3.0 and worst[2] <= 3.0
    # limited range holds up to the truncation error (yellow gives Cb = 15), so no sample byte is ever 0x00 -- which
    # is why the I_PCM payload never needs emulation-prevention bytes (7.4.1) although the encoder inserts none
    for px in imgs:
        _, Y, Cb, Cr, _ = oracle.encode_frame(px)
        assert Y.min() >= 16 and Y.max() <= 235 and Cb.min() >= 14 and Cb.max() <= 241 and Cr.min() >= 14 and Cr.max() <= 241


def test_mp4_samples_decode_with_the_spec_decoder(tor, oracle, tmp_path):
    w, h, frames, fps = 64, 48, 4, 30
    rng = np.random.default_rng(3)
    enc = [oracle.encode_frame(rng.uniform(0.0, 1.0, (h, w, 3))) for _ in range(frames)]
    stream = tor.h264_stream_header(w, h) + b"".join(e[4] for e in enc)
    src, dst = tmp_path / "a.264", tmp_path / "a.mp4"
    src.write_bytes(stream)
    assert tor.mp4_mux_file(str(src), str(dst), w, h, fps) == frames
    mp4 = dst.read_bytes()
    top = _boxes(mp4)
    moov = _boxes(mp4, *_child(mp4, top, "moov")[1:])
    trak = _boxes(mp4, *_child(mp4, moov, "trak")[1:])
    mdia = _boxes(mp4, *_child(mp4, trak, "mdia")[1:])
    minf = _boxes(mp4, *_child(mp4, mdia, "minf")[1:])
    stbl = _boxes(mp4, *_child(mp4, minf, "stbl")[1:])
    stsd = _child(mp4, stbl, "stsd")
    avc1 = _boxes(mp4, stsd[1] + 8, stsd[2])[0]
    avcc = _boxes(mp4, avc1[1] + 78, avc1[2])[0]
    c = mp4[avcc[1]:avcc[2]]
    # ISO/IEC 14496-15 5.2.4.1 AVCDecoderConfigurationRecord
    assert c[0] == 1 and (c[4] & 0xfc) == 0xfc and (c[5] & 0xe0) == 0xe0
    nal_len = (c[4] & 3) + 1
    pos, sets = 6, []
    for _ in range(c[5] & 0x1f):
        n = struct.unpack(">H", c[pos:pos + 2])[0]; sets.append(c[pos + 2:pos + 2 + n]); pos += 2 + n
    n_pps = c[pos]; pos += 1
    for _ in range(n_pps):
        n = struct.unpack(">H", c[pos:pos + 2])[0]; sets.append(c[pos + 2:pos + 2 + n]); pos += 2 + n
    assert pos == len(c) and c[1] == sets[0][1] and c[3] == sets[0][3]      # profile / level copied from the SPS
    stsz = _child(mp4, stbl, "stsz")
    const, count = struct.unpack(">II", mp4[stsz[1] + 4:stsz[1] + 12])
    sizes = [const] * count if const else list(struct.unpack(f">{count}I", mp4[stsz[1] + 12:stsz[1] + 12 + 4 * count]))
    offs_box = [b for b in stbl if b[0] in ("stco", "co64")][0]
    n_off = struct.unpack(">I", mp4[offs_box[1] + 4:offs_box[1] + 8])[0]
    fmt, width = (">%dQ", 8) if offs_box[0] == "co64" else (">%dI", 4)
    offs = struct.unpack(fmt % n_off, mp4[offs_box[1] + 8:offs_box[1] + 8 + width * n_off])
    assert count == n_off == frames
    rebuilt = b"".join(b"\x00\x00\x00\x01" + s for s in sets)
    for off, size in zip(offs, sizes):
        p = off
        while p < off + size:                                           # a sample = length-prefixed NAL units
            n = int.from_bytes(mp4[p:p + nal_len], "big")
            rebuilt += b"\x00\x00\x00\x01" + mp4[p + nal_len:p + nal_len + n]
            p += nal_len + n
        assert p == off + size
    sps, pps, pics = decode_stream(rebuilt)
    _check_parameter_sets(sps, pps, w, h)
    assert len(pics) == frames
    for (_, Y, Cb, Cr), e in zip(pics, enc):
        assert np.array_equal(Y, e[1]) and np.array_equal(Cb, e[2]) and np.array_equal(Cr, e[3])
    assert [n for _, n in split_annexb(rebuilt)] == [n for _, n in split_annexb(stream)]


@pytest.mark.gpu
def test_device_encoder_output_decodes_and_tracks_float_bt601(tor, oracle):
    """The fused device kernel's bytes through the spec decoder, on a rendered frame: planes == the kernel's planes,
    colours within the measured bound of float BT.601 computed from the DEVICE quantiser's RGB."""
    import torch
    ctx = tor.Context()
    cam, scene, _ = next(iter(tor.Animation(144, 256, 0.005, 0.2, 2.0).scenes(6)))
    ctx.upload(scene.list())
    h, w = 144, 256
    s = torch.cuda.current_stream().cuda_stream
    frame = torch.empty((h, w, 3), dtype=torch.float64, device="cuda")
    ctx.render_device(cam, h, w, 8, 2.2, 50, tor.make_options(seeding=tor.SEED_SAMPLE, accel=3), frame.data_ptr(), s)
    out = torch.zeros(tor.h264_frame_bytes(w, h), dtype=torch.uint8, device="cuda")
    Y = torch.zeros((h, w), dtype=torch.uint8, device="cuda")
    Cb = torch.zeros((h // 2, w // 2), dtype=torch.uint8, device="cuda"); Cr = torch.zeros_like(Cb)
    ctx.encode_frame_device(frame.data_ptr(), h, w, out.data_ptr(), Y.data_ptr(), Cb.data_ptr(), Cr.data_ptr(), s)
    rgb = torch.empty((h, w, 3), dtype=torch.uint8, device="cuda")
    ctx.quantize_rgb8_device(frame.data_ptr(), frame.numel(), rgb.data_ptr(), s)
    torch.cuda.synchronize()
    stream = tor.h264_stream_header(w, h) + out.cpu().numpy().tobytes() + out.cpu().numpy().tobytes()
    sps, pps, pics = decode_stream(stream)
    _check_parameter_sets(sps, pps, w, h)
    assert len(pics) == 2
    for _, dY, dCb, dCr in pics:
        assert np.array_equal(dY, Y.cpu().numpy()) and np.array_equal(dCb, Cb.cpu().numpy()) and np.array_equal(dCr, Cr.cpu().numpy())
    top_first = rgb.cpu().numpy()[::-1]                                   # video row 0 = top scanline = canvas row nrows-1
    e = _colour_errors(top_first, Y.cpu().numpy(), Cb.cpu().numpy(), Cr.cpu().numpy())
    print(f"device encoder vs float BT.601: max |dY| = {e[0]:.3f}, |dCb| = {e[1]:.3f}, |dCr| = {e[2]:.3f}")
    assert e[0] <= 2.1 and e[1] <= 3.0 and e[2] <= 3.0
    ctx.close()
