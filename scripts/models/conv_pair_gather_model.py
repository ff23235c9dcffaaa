#!/usr/bin/env python3
"""CPU model of qconv_mfma.hip's PAIR gather (two neighbouring output pixels per 4-byte load): for every (pixel pair, tap) it replays the kernel's
integer arithmetic - validity bits, the +2 / -2 byte realignment at the left / right border, the `& (va | vb)` clamp, the v_perm selector - on a
flat uint16 image and checks that (a) every load stays inside the tensor, (b) the two extracted elements are what im2col puts at (pixel, tap):
x[b, c, ih, iw] inside the image, 0 over the padding.  `check(...)` is imported by tests/test_qconv2d.py (CPU, no device)."""
import itertools

import numpy as np

IDENT, UP16, DOWN16, ZERO = 0x07060504, 0x05040C0C, 0x0C0C0706, 0x0C0C0C0C


def v_perm_b32(s0, s1, sel):
    """D.byte[i] = byte sel.byte[i] of {s0 (bytes 7..4), s1 (bytes 3..0)}; selector 0x0c = constant 0."""
    src = [(s1 >> (8 * i)) & 0xFF for i in range(4)] + [(s0 >> (8 * i)) & 0xFF for i in range(4)]
    out = 0
    for i in range(4):
        b = (sel >> (8 * i)) & 0xFF
        out |= (0 if b == 0x0C else src[b]) << (8 * i)
    return out


def check(B, C, H, W, KH, KW, sh, ph, pw, dh, dw, seed=0):
    sw = 1
    OH = (H + 2 * ph - dh * (KH - 1) - 1) // sh + 1
    OW = (W + 2 * pw - dw * (KW - 1) - 1) // sw + 1
    assert OW % 2 == 0 and W >= 2
    rng = np.random.default_rng(seed)
    x = rng.integers(1, 65535, size=(B, C, H, W), dtype=np.uint16)  # (never 0: a zero from the padding cannot hide behind a stored zero)
    flat = x.reshape(-1).view(np.uint8)
    nbytes = flat.size
    M = B * OH * OW
    for m in range(0, M, 2):
        b, l = divmod(m, OH * OW)
        oh, ow = divmod(l, OW)
        ih0, iw0 = oh * sh - ph, ow * sw - pw
        px_off = 2 * (b * C * H * W + oh * sh * W + ow * sw)
        for c, ki, kj in itertools.product(range(C), range(KH), range(KW)):
            ih, iw = ih0 + ki * dh, iw0 + kj * dw
            row_ok = 0 <= ih < H
            va = -1 if (row_ok and 0 <= iw < W) else 0
            vb = -1 if (row_ok and 0 <= iw + 1 < W) else 0
            off = 2 * ((c * H + ki * dh) * W + kj * dw - (ph * W + pw))
            addr = ((px_off + off + 2 * (va - vb)) & 0xFFFFFFFF) & ((va | vb) & 0xFFFFFFFF)
            sel = (va & IDENT) | (~va & UP16)
            sel = (vb & sel) | (~vb & DOWN16)
            sel = ((va | vb) & sel) | (~(va | vb) & ZERO)
            sel &= 0xFFFFFFFF
            assert addr + 4 <= nbytes, ("load leaves the tensor", m, c, ki, kj, addr, nbytes)
            word = int(flat[addr]) | int(flat[addr + 1]) << 8 | int(flat[addr + 2]) << 16 | int(flat[addr + 3]) << 24
            d = v_perm_b32(word, 0, sel)
            want_a = int(x[b, c, ih, iw]) if va else 0
            want_b = int(x[b, c, ih, iw + 1]) if vb else 0
            assert (d & 0xFFFF, d >> 16) == (want_a, want_b), ("wrong elements", m, c, ki, kj, hex(d), want_a, want_b, va, vb)
    return M // 2 * C * KH * KW


if __name__ == "__main__":
    n = 0
    for args in [(2, 2, 6, 8, 3, 3, 1, 1, 1, 1, 1), (1, 3, 5, 10, 3, 3, 2, 0, 2, 1, 1), (2, 1, 8, 14, 3, 5, 1, 1, 2, 1, 2), (3, 2, 5, 2, 3, 3, 1, 1, 1, 1, 1),
                 (1, 2, 6, 6, 1, 1, 1, 0, 0, 1, 1), (1, 1, 10, 12, 7, 7, 1, 3, 3, 1, 1), (2, 2, 4, 9, 1, 2, 1, 0, 0, 1, 1), (1, 1, 4, 4, 3, 3, 1, 2, 3, 2, 1)]:
        n += check(*args)
    print(f"pair gather model: {n} (pair, tap) loads in range and correct")
