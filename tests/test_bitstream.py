"""Syntax-side host code.  The first five tests re-express the reference's own
bit-reader known-answer tests (src/bitstream.rs:349-437) with the same vectors; the rest are derived by
hand from the cited reference code."""
import ctypes as C

import numpy as np
import pytest

from tmc2rs import _abi


@pytest.fixture(scope="module")
def lib():
    L = _abi.load_library()
    vp = C.c_void_p
    L.vpcc_bs_new.restype = vp
    L.vpcc_bs_new.argtypes = [C.c_char_p, C.c_size_t]
    L.vpcc_bs_free.argtypes = [vp]
    L.vpcc_bs_read.argtypes = [vp, C.c_uint, C.POINTER(C.c_uint32)]
    L.vpcc_bs_peek.argtypes = [vp, C.c_uint, C.POINTER(C.c_uint32)]
    L.vpcc_bs_read_uvlc.argtypes = [vp, C.POINTER(C.c_uint32)]
    L.vpcc_bs_read_svlc.argtypes = [vp, C.POINTER(C.c_int32)]
    L.vpcc_bs_byte_align.argtypes = [vp]
    L.vpcc_bs_reset.argtypes = [vp]
    L.vpcc_bs_copy_from.argtypes = [vp, vp, C.c_size_t, C.c_size_t]
    L.vpcc_bs_data.argtypes = [vp, C.POINTER(C.POINTER(C.c_uint8))]
    L.vpcc_bs_data.restype = C.c_size_t
    L.vpcc_bs_position.argtypes = [vp, C.POINTER(C.c_size_t), C.POINTER(C.c_uint)]
    return L


class BS:
    def __init__(self, lib, data):
        self.lib, self.h = lib, lib.vpcc_bs_new(bytes(data), len(data))

    def read(self, n):
        v = C.c_uint32()
        assert self.lib.vpcc_bs_read(self.h, n, C.byref(v)) == 0
        return v.value

    def peek(self, n):
        v = C.c_uint32()
        assert self.lib.vpcc_bs_peek(self.h, n, C.byref(v)) == 0
        return v.value

    def uvlc(self):
        v = C.c_uint32()
        assert self.lib.vpcc_bs_read_uvlc(self.h, C.byref(v)) == 0
        return v.value

    def svlc(self):
        v = C.c_int32()
        assert self.lib.vpcc_bs_read_svlc(self.h, C.byref(v)) == 0
        return v.value

    def data(self):
        p = C.POINTER(C.c_uint8)()
        n = self.lib.vpcc_bs_data(self.h, C.byref(p))
        return [p[i] for i in range(n)]

    def pos(self):
        b, s = C.c_size_t(), C.c_uint()
        self.lib.vpcc_bs_position(self.h, C.byref(b), C.byref(s))
        return b.value, s.value


# ---- the reference's five tests ------------------------------------------------------------------
def test_bitstream_read(lib):                         # src/bitstream.rs:349-360
    b = BS(lib, [0b10101010, 0b11110000, 0b11001001, 0b00110011])
    assert b.read(1) == 0b1
    assert b.read(3) == 0b010
    assert b.read(7) == 0b1010111
    assert b.read(11) == 0b10000110010
    assert b.read(4) == 0b0100
    assert b.read(6) == 0b110011
    lib.vpcc_bs_reset(b.h)
    assert b.read(8) == 0b10101010


def test_bitstream_peek(lib):                         # src/bitstream.rs:362-369
    b = BS(lib, [0b10101010])
    assert b.peek(1) == 1 and b.peek(1) == 1
    assert b.peek(3) == 0b101 and b.peek(3) == 0b101


UVLC = [0b10100110, 0b01000010, 0b10011000, 0b11100010, 0b00000100, 0b10001010, 0b00010110,
        0b00110000, 0b01101000, 0b11100001, 0b11100000]


def test_bitstream_read_uvlc(lib):                    # src/bitstream.rs:371-392
    b = BS(lib, UVLC)
    assert [b.uvlc() for _ in range(15)] == list(range(15))


def test_bitstream_read_svlc(lib):                    # src/bitstream.rs:394-415
    b = BS(lib, UVLC)
    assert [b.svlc() for _ in range(15)] == [0, 1, -1, 2, -2, 3, -3, 4, -4, 5, -5, 6, -6, 7, -7]


def test_copy_from(lib):                              # src/bitstream.rs:416-437
    b = BS(lib, [0b10101010, 0b11110000, 0b11001001, 0b00110011])
    b2 = BS(lib, [0b11001001, 0b00110011, 0b11001001, 0b11111111])
    assert lib.vpcc_bs_copy_from(b.h, b2.h, 1, 2) == 0
    assert b.data() == [0b00110011, 0b11001001, 0b11001001, 0b00110011]
    assert lib.vpcc_bs_copy_from(b.h, b2.h, 3, 1) == 0
    assert b.data() == [0b00110011, 0b11001001, 0b11111111, 0b00110011]
    assert lib.vpcc_bs_copy_from(b.h, b2.h, 0, 4) == 0
    assert b.data() == [0b00110011, 0b11001001, 0b11111111, 0b11001001, 0b00110011, 0b11001001, 0b11111111]


# ---- hand-derived from the cited code ------------------------------------------------------------
def test_byte_align_and_errors(lib):
    b = BS(lib, [0xFF, 0x0F])
    b.read(3)
    assert lib.vpcc_bs_byte_align(b.h) == 0 and b.pos() == (1, 0)     # stop bit read, then to the boundary
    assert lib.vpcc_bs_byte_align(b.h) == 0 and b.pos() == (2, 0)     # aligned: the wrinkle still eats one bit -> next byte
    v = C.c_uint32()
    assert lib.vpcc_bs_read(b.h, 1, C.byref(v)) != 0                   # past the end (the reference panics)
    assert lib.vpcc_bs_read(b.h, 33, C.byref(v)) != 0                  # bits > 32 (panic)


def test_v3c_sample_stream_split(lib):
    # header: precision_bytes_minus1 = 1 (u3) + 5 padding bits; units prefixed by 2-byte sizes
    u1 = bytes([0 << 3, 1, 2])                 # unit type 0 (VPS)
    u2 = bytes([1 << 3, 9, 9, 9, 9])           # unit type 1 (atlas data)
    u3 = bytes([4 << 3])                       # unit type 4
    stream = bytes([1 << 5]) + b"".join(len(u).to_bytes(2, "big") + u for u in (u1, u2, u3))
    types = (C.c_uint8 * 8)()
    offs = (C.c_size_t * 8)()
    sizes = (C.c_size_t * 8)()
    n, hs = C.c_uint32(), C.c_size_t()
    lib.vpcc_v3c_split.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.POINTER(C.c_uint32), C.POINTER(C.c_size_t)]
    assert lib.vpcc_v3c_split(stream, len(stream), 8, types, offs, sizes, C.byref(n), C.byref(hs)) == 0
    assert n.value == 3 and list(types[:3]) == [0, 1, 4] and list(sizes[:3]) == [3, 5, 1]
    assert [stream[offs[i]:offs[i] + sizes[i]] for i in range(3)] == [u1, u2, u3]
    assert hs.value == 1 + 3 * 2                 # as the reference accounts it (reader.rs:631-638)
    assert lib.vpcc_v3c_split(stream[:-1], len(stream) - 1, 8, types, offs, sizes, C.byref(n), C.byref(hs)) != 0   # truncated


def test_sample_stream_to_bytestream_hevc(lib):
    # NAL types: VPS(32) SPS(33) PPS(34) get 4-byte start codes, a slice (type 1) after them a 3-byte one
    def nal(t, payload):
        return bytes([(t << 1) & 0x7E, 1]) + payload
    nals = [nal(32, b"\xAA"), nal(33, b"\xBB\xBB"), nal(34, b"\xCC"), nal(1, b"\xDD\xDD\xDD"), nal(1, b"\xEE")]
    ss = b"".join(len(x).to_bytes(4, "big") + x for x in nals)
    out = (C.c_uint8 * 256)()
    n = C.c_size_t()
    lib.vpcc_sample_stream_to_bytestream.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t,
                                                     C.POINTER(C.c_size_t)]
    assert lib.vpcc_sample_stream_to_bytestream(ss, len(ss), 1, out, 256, C.byref(n)) == 0
    got = bytes(out[:n.value])
    # first NAL always 4-byte; next code length is decided by the NEXT NAL's type: 33, 34 -> long; slices -> short
    exp = b"\0\0\0\1" + nals[0] + b"\0\0\0\1" + nals[1] + b"\0\0\0\1" + nals[2] + b"\0\0\1" + nals[3] + b"\0\0\1" + nals[4]
    assert got == exp
    assert lib.vpcc_sample_stream_to_bytestream(ss, len(ss), 0, out, 256, C.byref(n)) == 0   # H264: always long
    assert bytes(out[:n.value]).count(b"\0\0\0\1") == 5
    assert lib.vpcc_sample_stream_to_bytestream(ss[:-1], len(ss) - 1, 1, out, 256, C.byref(n)) != 0


def test_patch_from_intra_pdu(lib):
    class FP(C.Structure):
        _fields_ = [(n, C.c_uint32) for n in ("log2_patch_packing_block_size", "geometry_3d_bitdepth", "pos_min_d_quantizer",
                                              "patch_size_quantizer_present_flag", "patch_size_info_quantizer_x",
                                              "patch_size_info_quantizer_y", "plr_enabled_flag", "reserved")]

    class PDU(C.Structure):
        _fields_ = [(n, C.c_uint32) for n in ("pos_2d_x", "pos_2d_y", "size_2d_x_minus1", "size_2d_y_minus1", "pos_3d_offset_u",
                                              "pos_3d_offset_v", "pos_3d_offset_d", "pos_3d_range_d", "projection_id",
                                              "orientation_index", "lod_enabled_flag", "reserved")]
    lib.vpcc_patch_from_intra_pdu.argtypes = [C.POINTER(FP), C.POINTER(PDU), C.POINTER(_abi.Patch)]
    fp = FP(4, 10, 2, 0, 0, 0, 0, 0)               # block 16, 10-bit geometry, minLevel = 4
    out = _abi.Patch()
    pdu = PDU(3, 5, 6, 1, 100, 200, 7, 0, 1, 1, 0, 0)     # projection 1 -> axes (1,2,0) mode 0; Swap
    assert lib.vpcc_patch_from_intra_pdu(C.byref(fp), C.byref(pdu), C.byref(out)) == 0
    assert (out.u0, out.v0, out.size_u0, out.size_v0, out.u1, out.v1) == (3, 5, 7, 2, 100, 200)
    assert (out.normal_axis, out.tangent_axis, out.bitangent_axis, out.projection_mode) == (1, 2, 0, 0)
    assert out.d1 == 7 * 4 and out.orientation == 1 and out.lod_x == out.lod_y == 1 and out.axis_of_additional_plane == 0
    pdu.projection_id = 5                                  # axes (2,0,1), mode 1: d1 = 2^10 - 7*4
    assert lib.vpcc_patch_from_intra_pdu(C.byref(fp), C.byref(pdu), C.byref(out)) == 0
    assert (out.normal_axis, out.tangent_axis, out.bitangent_axis, out.projection_mode, out.d1) == (2, 0, 1, 1, 1024 - 28)
    fp.patch_size_quantizer_present_flag, fp.patch_size_info_quantizer_x, fp.patch_size_info_quantizer_y = 1, 3, 2
    assert lib.vpcc_patch_from_intra_pdu(C.byref(fp), C.byref(pdu), C.byref(out)) == 0
    assert (out.size_u0, out.size_v0) == (4, 1)            # ceil(7*8/16) = 4, ceil(2*4/16) = 1 (decoder.rs:442-452)
    pdu.projection_id = 6                                  # 45-degree plane: the hot path rejects it later
    assert lib.vpcc_patch_from_intra_pdu(C.byref(fp), C.byref(pdu), C.byref(out)) == 0 and out.axis_of_additional_plane == 1
    pdu.lod_enabled_flag = 1
    assert lib.vpcc_patch_from_intra_pdu(C.byref(fp), C.byref(pdu), C.byref(out)) == _abi.VPCC_ERR_UNSUPPORTED
