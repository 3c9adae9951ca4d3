"""A small H.264 DECODER for the only kind of stream the animation driver writes (baseline, CAVLC, every
macroblock I_PCM) -- TEST INFRASTRUCTURE, written from the syntax tables of ITU-T Rec. H.264 (04/2017), not from
the encoder in this repository or the reference's `io/h264.nim`:

    Annex B byte stream (B.1)                -> NAL units, start codes, emulation-prevention bytes (7.4.1)
    seq_parameter_set_rbsp     (7.3.2.1.1)   -> every field up to the VUI flag, rbsp_trailing_bits
    pic_parameter_set_rbsp     (7.3.2.2)     -> every field, rbsp_trailing_bits
    slice_layer_without_partitioning_rbsp    -> slice_header (7.3.3) as far as an IDR I slice needs it, slice_data
    (7.3.2.8 / 7.3.4), macroblock_layer (7.3.5): mb_type ue(v) (Table 7-11: 25 = I_PCM), pcm_alignment_zero_bit,
    256 + 2 * 64 pcm samples, then rbsp_slice_trailing_bits.

No player, demuxer or third-party parser exists in this image (no network: PyAV / ffprobe cannot be installed), so
this is the closest thing to "somebody else's decoder" available: it derives every field position from the
standard and would reject a stream whose SPS/PPS/slice header or macroblock framing a real decoder would reject.
It is NOT a pin to the reference (which holds no encoded frame).
"""
from __future__ import annotations

import numpy as np


class BitReader:
    def __init__(self, data: bytes):
        self.d = data
        self.pos = 0  # bit position

    def u(self, n: int) -> int:
        v = 0
        for _ in range(n):
            byte = self.d[self.pos >> 3]
            v = (v << 1) | ((byte >> (7 - (self.pos & 7))) & 1)
            self.pos += 1
        return v

    def ue(self) -> int:  # 9.1 Exp-Golomb
        zeros = 0
        while self.u(1) == 0:
            zeros += 1
            assert zeros < 32
        return (1 << zeros) - 1 + (self.u(zeros) if zeros else 0)

    def se(self) -> int:
        k = self.ue()
        return (k + 1) // 2 if k & 1 else -(k // 2)

    def byte_aligned(self) -> bool:
        return (self.pos & 7) == 0

    def more_rbsp_data(self) -> bool:  # 7.2: is there anything before the rbsp_trailing_bits?
        last = len(self.d) - 1
        while last >= 0 and self.d[last] == 0:
            last -= 1
        assert last >= 0
        stop_bit = last * 8 + (7 - ((self.d[last] & -self.d[last]).bit_length() - 1))
        return self.pos < stop_bit

    def trailing_bits(self):  # 7.3.2.11
        assert self.u(1) == 1, "rbsp_stop_one_bit"
        while not self.byte_aligned():
            assert self.u(1) == 0, "rbsp_alignment_zero_bit"


def split_annexb(stream: bytes):
    """B.1.1: returns [(start_code_length, nal_bytes_with_emulation_prevention)]."""
    n = len(stream)
    starts = []
    i = 0
    while i + 3 <= n:
        if stream[i] == 0 and stream[i + 1] == 0 and stream[i + 2] == 1:
            starts.append(i)
            i += 3
        else:
            i += 1
    assert starts and all(b == 0 for b in stream[:starts[0]]), "leading_zero_8bits only before the first start code"
    out = []
    for k, s in enumerate(starts):
        e = starts[k + 1] if k + 1 < len(starts) else n
        body = stream[s + 3:e]
        # trailing_zero_8bits / the zero_byte of the next 4-byte start code belong to the stream, not to the NAL unit
        sc_len = 4 if s > 0 and stream[s - 1] == 0 else 3
        if k + 1 < len(starts):
            while body and body[-1] == 0:
                body = body[:-1]
        out.append((sc_len, bytes(body)))
    return out


def unescape(nal: bytes) -> bytes:
    """7.3.1 / 7.4.1: drop emulation_prevention_three_byte; reject forbidden byte patterns."""
    out = bytearray()
    zeros = 0
    i = 0
    while i < len(nal):
        b = nal[i]
        if zeros >= 2 and b == 3:
            assert i + 1 == len(nal) or nal[i + 1] <= 3, "0x000003 must be followed by 00..03"
            zeros = 0
            i += 1
            continue
        assert not (zeros >= 2 and b < 3), "0x000000 / 0x000001 / 0x000002 inside a NAL unit"
        out.append(b)
        zeros = zeros + 1 if b == 0 else 0
        i += 1
    return bytes(out)


def parse_sps(rbsp: bytes) -> dict:
    r = BitReader(rbsp)
    s = {"profile_idc": r.u(8)}
    s["constraint_flags"] = r.u(6)
    assert r.u(2) == 0, "reserved_zero_2bits"
    s["level_idc"] = r.u(8)
    s["sps_id"] = r.ue()
    assert s["profile_idc"] not in (100, 110, 122, 244, 44, 83, 86, 118, 128, 138, 139, 134, 135), "high profiles carry chroma_format_idc"
    s["log2_max_frame_num"] = r.ue() + 4
    s["poc_type"] = r.ue()
    if s["poc_type"] == 0:
        s["log2_max_poc_lsb"] = r.ue() + 4
    elif s["poc_type"] == 1:
        raise AssertionError("pic_order_cnt_type 1 not expected")
    s["max_num_ref_frames"] = r.ue()
    s["gaps_in_frame_num_allowed"] = r.u(1)
    s["pic_width_in_mbs"] = r.ue() + 1
    s["pic_height_in_map_units"] = r.ue() + 1
    s["frame_mbs_only"] = r.u(1)
    if not s["frame_mbs_only"]:
        s["mb_adaptive_frame_field"] = r.u(1)
    s["direct_8x8_inference"] = r.u(1)
    s["frame_cropping"] = r.u(1)
    if s["frame_cropping"]:
        s["crop"] = [r.ue() for _ in range(4)]
    s["vui_present"] = r.u(1)
    assert not s["vui_present"]
    r.trailing_bits()
    assert r.pos == len(rbsp) * 8, "bytes after rbsp_trailing_bits"
    return s


def parse_pps(rbsp: bytes) -> dict:
    r = BitReader(rbsp)
    p = {"pps_id": r.ue(), "sps_id": r.ue(), "entropy_coding_mode": r.u(1), "bottom_field_pic_order": r.u(1)}
    p["num_slice_groups"] = r.ue() + 1
    assert p["num_slice_groups"] == 1
    p["num_ref_idx_l0"] = r.ue() + 1
    p["num_ref_idx_l1"] = r.ue() + 1
    p["weighted_pred"] = r.u(1)
    p["weighted_bipred_idc"] = r.u(2)
    p["pic_init_qp"] = r.se() + 26
    p["pic_init_qs"] = r.se() + 26
    p["chroma_qp_index_offset"] = r.se()
    p["deblocking_filter_control_present"] = r.u(1)
    p["constrained_intra_pred"] = r.u(1)
    p["redundant_pic_cnt_present"] = r.u(1)
    assert not r.more_rbsp_data(), "transform_8x8_mode etc. not expected in a baseline PPS"
    r.trailing_bits()
    assert r.pos == len(rbsp) * 8
    return p


def decode_idr_ipcm_slice(nal_header: int, rbsp: bytes, sps: dict, pps: dict):
    """slice_layer_without_partitioning_rbsp of an IDR picture whose macroblocks are all I_PCM -> (Y, Cb, Cr)."""
    nal_ref_idc, nal_unit_type = (nal_header >> 5) & 3, nal_header & 0x1f
    assert nal_unit_type == 5
    # 7.4.1: "nal_ref_idc shall not be equal to 0 for NAL units with nal_unit_type equal to 5".  The reference's
    # constant slice header (io/h264.nim:38, bytes 00 00 00 01 05 ...) writes nal_ref_idc = 0 -- the one place where
    # its stream departs from the letter of the standard (players accept it: nothing ever references the picture).
    # The product reproduces the reference's bytes, so the decoder records the quirk instead of rejecting the stream.
    quirks = ["idr_with_nal_ref_idc_0"] if nal_ref_idc == 0 else []
    r = BitReader(rbsp)
    h = {"first_mb_in_slice": r.ue(), "slice_type": r.ue(), "pps_id": r.ue(), "quirks": quirks}
    assert h["first_mb_in_slice"] == 0 and h["slice_type"] % 5 == 2, "one I slice per picture"
    assert h["pps_id"] == pps["pps_id"]
    h["frame_num"] = r.u(sps["log2_max_frame_num"])
    assert h["frame_num"] == 0, "frame_num of an IDR picture (7.4.3)"
    assert sps["frame_mbs_only"]
    h["idr_pic_id"] = r.ue()
    if sps["poc_type"] == 0:
        h["poc_lsb"] = r.u(sps["log2_max_poc_lsb"])
        assert not pps["bottom_field_pic_order"]
    assert not pps["redundant_pic_cnt_present"]
    # I slice: no ref_pic_list_modification, no pred_weight_table
    if nal_ref_idc != 0:                       # 7.3.3: dec_ref_pic_marking() only for reference pictures
        h["no_output_of_prior_pics"] = r.u(1)  # (7.3.3.3, IDR branch)
        h["long_term_reference"] = r.u(1)
    assert not pps["entropy_coding_mode"], "CAVLC expected (mb_type as ue(v))"
    h["slice_qp_delta"] = r.se()
    if pps["deblocking_filter_control_present"]:
        h["disable_deblocking_filter_idc"] = r.ue()
        if h["disable_deblocking_filter_idc"] != 1:
            r.se(); r.se()
    # slice_data (7.3.4), CAVLC, I slice: no mb_skip_run
    wmb, hmb = sps["pic_width_in_mbs"], sps["pic_height_in_map_units"]
    Y = np.zeros((hmb * 16, wmb * 16), np.uint8)
    Cb = np.zeros((hmb * 8, wmb * 8), np.uint8)
    Cr = np.zeros_like(Cb)
    raw = np.frombuffer(rbsp, np.uint8)
    for mb in range(wmb * hmb):
        mb_type = r.ue()
        assert mb_type == 25, f"macroblock {mb}: mb_type {mb_type}, expected 25 (I_PCM, Table 7-11)"
        while not r.byte_aligned():
            assert r.u(1) == 0, "pcm_alignment_zero_bit"
        k = r.pos >> 3
        i, j = divmod(mb, wmb)
        Y[i * 16:(i + 1) * 16, j * 16:(j + 1) * 16] = raw[k:k + 256].reshape(16, 16)
        Cb[i * 8:(i + 1) * 8, j * 8:(j + 1) * 8] = raw[k + 256:k + 320].reshape(8, 8)
        Cr[i * 8:(i + 1) * 8, j * 8:(j + 1) * 8] = raw[k + 320:k + 384].reshape(8, 8)
        r.pos += 384 * 8
        assert r.more_rbsp_data() == (mb + 1 < wmb * hmb), "more_rbsp_data() must end the slice exactly after the last macroblock"
    r.trailing_bits()                           # rbsp_slice_trailing_bits (CAVLC: no cabac_zero_words)
    assert r.pos == len(rbsp) * 8
    return h, Y, Cb, Cr


def decode_stream(stream: bytes):
    """Annex-B stream -> (sps, pps, [(slice_header, Y, Cb, Cr) per picture])."""
    sps = pps = None
    pictures = []
    for sc_len, nal in split_annexb(stream):
        assert nal, "empty NAL unit"
        assert (nal[0] & 0x80) == 0, "forbidden_zero_bit"
        t = nal[0] & 0x1f
        assert unescape(nal) == nal or t in (7, 8), "sample data must not need emulation prevention (limited-range BT.601 has no zero bytes)"
        rbsp = unescape(nal)[1:]
        if t == 7:
            assert sc_len == 4, "B.1.2: zero_byte is required in front of an SPS"
            sps = parse_sps(rbsp)
        elif t == 8:
            assert sc_len == 4, "B.1.2: zero_byte is required in front of a PPS"
            pps = parse_pps(rbsp)
            assert sps is not None and pps["sps_id"] == sps["sps_id"]
        elif t == 5:
            assert sps is not None and pps is not None, "slice before its parameter sets"
            assert sc_len == 4, "B.1.2: zero_byte is required for the first NAL unit of an access unit"
            h, Y, Cb, Cr = decode_idr_ipcm_slice(nal[0], rbsp, sps, pps)
            if sps.get("frame_cropping"):   # 7.4.2.1.1, frame_mbs_only, 4:2:0: CropUnitX = CropUnitY = 2 luma samples
                l, r_, t, b = sps["crop"]
                Y = Y[2 * t: Y.shape[0] - 2 * b, 2 * l: Y.shape[1] - 2 * r_]
                Cb = Cb[t: Cb.shape[0] - b, l: Cb.shape[1] - r_]
                Cr = Cr[t: Cr.shape[0] - b, l: Cr.shape[1] - r_]
            pictures.append((h, Y, Cb, Cr))
        else:
            raise AssertionError(f"unexpected nal_unit_type {t}")
    return sps, pps, pictures


def bt601_limited_float(rgb8: np.ndarray):
    """ITU-R BT.601 R'G'B' (0..255, full range) -> Y'CbCr limited range, 4:2:0 by averaging 2x2 blocks; floats."""
    r, g, b = (rgb8[..., k].astype(np.float64) for k in range(3))
    y = 16.0 + (65.481 * r + 128.553 * g + 24.966 * b) / 255.0
    cb = 128.0 + (-37.797 * r - 74.203 * g + 112.0 * b) / 255.0
    cr = 128.0 + (112.0 * r - 93.786 * g - 18.214 * b) / 255.0
    def sub(p):
        return (p[0::2, 0::2] + p[0::2, 1::2] + p[1::2, 0::2] + p[1::2, 1::2]) / 4.0
    return y, sub(cb), sub(cr)
