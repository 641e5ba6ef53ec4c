"""Display epilogue (dumpRGB h4m:897-926) over ALL 2^24 (Y, U, V) triples.

The kernel clamps and truncates with v_floor_f32 + v_cvt_pk_u8_f32 (saturating pack, round to nearest -- exact behind a floor;
tools/ubench/cvt_probe.hip) instead of two compares, two selects and a convert.  CPU: that arithmetic, restated in numpy
with one float32 rounding per operation, equals the reference's own dumpRGB (oracle/_ref) and the oracle's restatement on a
4096x4096 picture that holds every triple.  GPU: the kernel itself converts the same picture, tolerance 0."""
import numpy as np
import pytest

from oracle import bridge

W = H = 4096


def all_triples_picture() -> np.ndarray:
    """4:2:0 picture in which every (Y, U, V) occurs: chroma sample c holds pair c % 65536 (U = low byte, V = high byte),
    its four luma samples hold Y = 4 * (c // 65536) + k"""
    cw, ch = W // 2, H // 2
    c = np.arange(cw * ch, dtype=np.uint32).reshape(ch, cw)
    u = (c & 0xFF).astype(np.uint8)
    v = ((c >> 8) & 0xFF).astype(np.uint8)
    rep = (c >> 16).astype(np.uint32)                     # 0..63
    y = np.empty((H, W), dtype=np.uint8)
    for k, (dy, dx) in enumerate(((0, 0), (0, 1), (1, 0), (1, 1))):
        y[dy::2, dx::2] = (4 * rep + k).astype(np.uint8)
    return np.concatenate([y.ravel(), u.ravel(), v.ravel()])


def kernel_arithmetic(yuv: np.ndarray) -> np.ndarray:
    """the kernel's float path: one rounding per operation, then floor and a saturating pack"""
    f32 = np.float32
    y = yuv[:W * H].reshape(H, W).astype(f32)
    u = np.repeat(np.repeat(yuv[W * H:W * H + W * H // 4].reshape(H // 2, W // 2), 2, 0), 2, 1).astype(f32) - f32(128)
    v = np.repeat(np.repeat(yuv[W * H + W * H // 4:].reshape(H // 2, W // 2), 2, 0), 2, 1).astype(f32) - f32(128)
    r = y + f32(1.402) * v
    g = (y - f32(0.34414) * u) - f32(0.71414) * v
    b = y + f32(1.772) * u
    out = np.empty((H, W, 3), dtype=np.uint8)
    for i, ch in enumerate((r, g, b)):
        assert ch.dtype == np.float32
        out[..., i] = np.clip(np.floor(ch), 0, 255).astype(np.uint8)          # v_floor_f32; v_cvt_pk_u8_f32 saturates
    return out


def test_every_triple_occurs():
    yuv = all_triples_picture()
    y = yuv[:W * H].reshape(H, W)[::2, ::2].astype(np.uint32)
    u = yuv[W * H:W * H + W * H // 4].astype(np.uint32)
    v = yuv[W * H + W * H // 4:].astype(np.uint32)
    seen = np.zeros(1 << 24, dtype=bool)
    for k, (dy, dx) in enumerate(((0, 0), (0, 1), (1, 0), (1, 1))):
        yy = yuv[:W * H].reshape(H, W)[dy::2, dx::2].ravel().astype(np.uint32)
        seen[(yy << 16) | (u << 8) | v] = True
    assert seen.all() and y.size == 1 << 22


def test_floor_and_saturating_pack_equal_the_reference_dumprgb_on_all_triples():
    yuv = all_triples_picture()
    mine = kernel_arithmetic(yuv)
    assert np.array_equal(mine.reshape(-1), bridge.oracle_rgb(yuv, W, H).reshape(-1))
    if bridge.have_ref():                                  # the unmodified reference, where /root/reference exists
        assert np.array_equal(mine.reshape(-1), bridge.ref_rgb(yuv, W, H).reshape(-1))


@pytest.mark.gpu
def test_gpu_rgb_kernel_on_all_triples(gpu_ctx):
    yuv = all_triples_picture()
    got = gpu_ctx.convert_yuv420_rgb(yuv, W, H)
    assert np.array_equal(got.reshape(-1), bridge.oracle_rgb(yuv, W, H).reshape(-1))
