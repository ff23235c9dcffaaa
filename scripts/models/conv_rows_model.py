"""CPU model of the ROW form of the implicit-GEMM convolution (csrc/qconv_mfma.hip, qconv2d_rows_kernel), index for index.

What it follows: the lane's pixel pair (clamped 8-byte window, padding bits, v_perm selectors), the scalar row arithmetic (magic-number division,
"behind the last row" through bit 31), the range-checked buffer loads (an offset with the top bits set reads zeros), the 4 x 4 transposition by
v_perm_b32, the LDS image ([row][224 bytes], 16-byte slot c of row r at c ^ (r >> 2 & 3), tap blocks of 64 bytes), the weight bytes' (row, tap) ->
[tap][row] regrouping, the MFMA fragment addresses and the accumulator -> output mapping, the K split.  What it cannot show is the hardware's side
of those instructions (2-byte aligned 8-byte buffer loads, the range check) - that is the GPU parity test's job.

    python scripts/models/conv_rows_model.py          # a handful of geometries against a direct convolution, and the LDS bank model
"""
import itertools

import numpy as np

BM, BN, NT, RT, RS = 128, 128, 512, 32, 224
OP_BYTES = BM * RS

GROUPS_B128 = [
    list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
    list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
    list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)),
    list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64)),
]


def div_magic(d):
    return 0 if d <= 1 else ((1 << 32) + d - 1) // d


def perm(src0, src1, sel):
    """v_perm_b32 D = perm({src0, src1}, sel): selector byte 0..3 -> src1's bytes, 4..7 -> src0's, 0x0c -> zero."""
    pool = [(src1 >> (8 * i)) & 0xFF for i in range(4)] + [(src0 >> (8 * i)) & 0xFF for i in range(4)]
    out = 0
    for i in range(4):
        s = (sel >> (8 * i)) & 0xFF
        assert s <= 7 or s == 0x0C, hex(sel)
        out |= (0 if s == 0x0C else pool[s]) << (8 * i)
    return out


def rows_eligible(cin, KH, KW, W, OW, sw, dw):
    return KW == 3 and dw == 1 and W >= 4 and KH <= 31 and (cin * KH) % 8 == 0


def rows_pairs(OW, sw):
    """pixel pairs (one 8-byte window for two neighbouring output pixels) need stride 1 along the width and an even OW; else one pixel per thread"""
    return sw == 1 and OW % 2 == 0


def conv_rows_model(x16, wq, B, cin, H, W, OC, KH, OH, OW, sh, ph, pw, dh, S=1, sw=1, single=None):
    """x16: uint16 [B, cin, H, W] (any 16-bit payload), wq: int8 [OC, cin*KH*3].  Returns int64 [B, OC, OH, OW] sums of payload * weight with
    the payload read as a signed 16-bit integer (exact arithmetic: what matters is WHICH elements meet)."""
    KW = 3
    M, N, K, R = B * OH * OW, OC, cin * KH * KW, cin * KH
    assert rows_eligible(cin, KH, KW, W, OW, sw, 1)
    if single is None:
        single = not rows_pairs(OW, sw)
    assert single or rows_pairs(OW, sw)
    NL = 128 if single else 64   # threads with distinct pixel state
    NRW = 8 if single else 4     # window rows per thread
    xbytes = x16.reshape(-1).view(np.uint8)
    xlen = xbytes.size
    wbytes = wq.reshape(-1).view(np.uint8)
    kh_magic = div_magic(KH)
    nk_all = (R + RT - 1) // RT
    S = min(S, nk_all)
    out = np.zeros((B, OC, OH, OW), dtype=np.int64)
    L = OH * OW

    def buf_load8(buf, nbytes, off):
        off &= 0xFFFFFFFF
        if off + 8 > nbytes:  # range check (the kernel only relies on it for offsets with the top bits set)
            assert off >= 0x80000000, ("an in-range row must be wholly inside the tensor", off, nbytes)
            return 0, 0
        d = buf[off:off + 8]
        return int.from_bytes(d[:4].tobytes(), "little"), int.from_bytes(d[4:].tobytes(), "little")

    mtiles, ntiles = (M + BM - 1) // BM, (N + BN - 1) // BN
    for mt, nt, sp in itertools.product(range(mtiles), range(ntiles), range(S)):
        m0 = mt * BM
        kt_lo = sp * nk_all // S
        nk = (sp + 1) * nk_all // S - kt_lo
        assert nk >= 1
        acc = np.zeros((8, 4, 2, 64, 4), dtype=np.int64)  # [wave][i][j][lane][r]
        # per-lane pixel-pair state (the same in every wave)
        px_base, nok, selN0, selN1 = [], [], [], []
        for lane in range(NL):  # (single: tid & 127)
            m = m0 + lane if single else m0 + 2 * lane
            m = m if m < M else (M - 1 if single else M - 2)
            b, l = divmod(m, L)
            oh, ow = divmod(l, OW)
            iw0 = ow * sw - pw
            ws = min(max(iw0, 0), W - 4)
            delta = iw0 - ws
            px_base.append(2 * ((b * cin * H + (oh * sh - ph)) * W + ws))
            nk_bits = 0x80000000
            for i in range(KH):
                ih = oh * sh - ph + i * dh
                if ih < 0 or ih >= H:
                    nk_bits |= 1 << i
            nok.append(nk_bits)

            def half_sel(j):
                iw, q = iw0 + j, j + delta
                if 0 <= iw < W:
                    assert 0 <= q <= 3
                    return ((2 * q + 1) << 8) | (2 * q)
                return 0x0C0C

            selN0.append(half_sel(0) | (half_sel(1) << 16))
            selN1.append(half_sel(2) | ((0x0C0C if single else half_sel(3)) << 16))

        for t in range(nk):
            T = kt_lo + t
            lds = np.zeros(2 * OP_BYTES, dtype=np.uint8)
            written = np.zeros(2 * OP_BYTES, dtype=bool)

            def st(addr, value, nbytes):
                assert addr % 8 == 0 and not written[addr:addr + nbytes].any()
                lds[addr:addr + nbytes] = np.frombuffer(int(value).to_bytes(nbytes, "little"), dtype=np.uint8)
                written[addr:addr + nbytes] = True

            for wave in range(8):
                # scalar row arithmetic
                roffs, bits = [], []
                for u in range(NRW):
                    r = T * RT + (8 * (wave >> 1) if single else 4 * wave) + u
                    c = ((r * kh_magic) >> 32) + (0 if kh_magic else r)
                    i = r - c * KH
                    assert r >= R or (0 <= i < KH and c < cin)
                    roffs.append((2 * ((c * H + i * dh) * W)) & 0xFFFFFFFF)
                    bits.append(i if r < R else 31)
                for lane in range(64):
                    tid = wave * 64 + lane
                    ps = (tid & 127) if single else lane  # index of the thread's pixel state
                    D = []
                    for u in range(NRW):
                        pad = 0xFFFFFFFF if (nok[ps] >> (bits[u] & 31)) & 1 else 0
                        addr = ((px_base[ps] + roffs[u]) & 0xFFFFFFFF) | pad
                        D.append(buf_load8(xbytes, xlen, addr))
                    n0 = [perm(D[u][1], D[u][0], selN0[ps]) for u in range(NRW)]
                    n1 = [perm(D[u][1], D[u][0], selN1[ps]) for u in range(NRW)]
                    LO, HI = 0x05040100, 0x07060302
                    if single:  # tap j of the eight rows: 16 bytes of tap block j, LDS row tid & 127, slot tid >> 7 of the block
                        awr = (tid & 127) * RS + (((tid >> 7) ^ ((tid >> 2) & 3)) << 4)
                        taps = [[perm(n0[2 * h + 1], n0[2 * h], LO) for h in range(4)], [perm(n0[2 * h + 1], n0[2 * h], HI) for h in range(4)],
                                [perm(n1[2 * h + 1], n1[2 * h], LO) for h in range(4)]]
                        for j in range(3):
                            v = 0
                            for h in range(4):
                                v |= taps[j][h] << (32 * h)
                            st(awr + 64 * j, v, 16)
                    p = [] if single else [
                        (perm(n0[1], n0[0], LO), perm(n0[3], n0[2], LO)),
                        (perm(n0[1], n0[0], HI), perm(n0[3], n0[2], HI)),
                        (perm(n1[1], n1[0], LO), perm(n1[3], n1[2], LO)),
                        (perm(n1[1], n1[0], HI), perm(n1[3], n1[2], HI)),
                    ]
                    awr = (2 * lane) * RS + (((wave >> 1) ^ ((lane >> 1) & 3)) << 4) + (wave & 1) * 8
                    for px in range(0 if single else 2):
                        for j in range(3):
                            lo, hi = p[j + px]
                            st(awr + px * RS + 64 * j, lo | (hi << 32), 8)
                    # weights
                    n = min(nt * BN + (tid >> 2), N - 1)
                    part = tid & 3
                    woff = n * K + 24 * part
                    krem = K - T * 96
                    wb = []
                    for j in range(3):
                        off = woff + T * 96 + 8 * j if 24 * part + 8 * j < krem else 0xFFFFFFFF
                        lo, hi = buf_load8(wbytes, N * K, off)
                        wb += [lo, hi]
                    bwr = OP_BYTES + (tid >> 2) * RS + (((tid & 3) ^ ((tid >> 4) & 3)) << 4)
                    for j in range(3):
                        v = 0
                        for rl in range(8):
                            e = 3 * rl + j
                            byte = (wb[e >> 2] >> (8 * (e & 3))) & 0xFF
                            sval = byte - 256 if byte >= 128 else byte  # "converted" weight: the int8 value as a 16-bit integer
                            v |= (sval & 0xFFFF) << (16 * rl)
                        st(bwr + 64 * j, v, 16)
            # every byte the fragments read was written exactly once (the padding bytes 192..223 of a row stay untouched)
            img = lds.view(np.int16)
            for wave in range(8):
                wm, wn = wave >> 2, wave & 3
                for kk in range(3):
                    fa = np.zeros((4, 64, 8), dtype=np.int64)
                    fb = np.zeros((2, 64, 8), dtype=np.int64)
                    for lane in range(64):
                        slot = ((lane >> 4) ^ ((lane >> 2) & 3)) << 4
                        for i in range(4):
                            a0 = (wm * 64 + (lane & 15)) * RS + slot + i * 16 * RS + kk * 64
                            assert written[a0:a0 + 16].all()
                            fa[i, lane] = img[a0 // 2:a0 // 2 + 8]
                        for j in range(2):
                            b0 = OP_BYTES + (wn * 32 + (lane & 15)) * RS + slot + j * 16 * RS + kk * 64
                            assert written[b0:b0 + 16].all()
                            fb[j, lane] = img[b0 // 2:b0 // 2 + 8]
                    for i in range(4):
                        A = fa[i].reshape(4, 16, 8).transpose(1, 0, 2).reshape(16, 32)  # [row][k]: lane = 16 g + row holds k = 8 g ..
                        for j in range(2):
                            Bm = fb[j].reshape(4, 16, 8).transpose(1, 0, 2).reshape(16, 32)
                            Dm = A @ Bm.T  # [row][col]
                            for lane in range(64):
                                for r in range(4):
                                    acc[wave, i, j, lane, r] += Dm[(lane >> 4) * 4 + r, lane & 15]
        # accumulators -> output (store_tile's mapping); the K split adds partial tiles
        for wave in range(8):
            wm, wn = wave >> 2, wave & 3
            for i, j, lane, r in itertools.product(range(4), range(2), range(64), range(4)):
                m = m0 + wm * 64 + i * 16 + (lane >> 4) * 4 + r
                n = nt * BN + wn * 32 + j * 16 + (lane & 15)
                if m < M and n < N:
                    b, l = divmod(m, L)
                    out[b, n, l // OW, l % OW] += acc[wave, i, j, lane, r]
    return out


def direct(x16, wq, B, cin, H, W, OC, KH, OH, OW, sh, ph, pw, dh, sw=1):
    xs = x16.view(np.int16).astype(np.int64)
    w = wq.astype(np.int64).reshape(OC, cin, KH, 3)
    out = np.zeros((B, OC, OH, OW), dtype=np.int64)
    xp = np.zeros((B, cin, H + 2 * ph + 64, W + 2 * pw + 64), dtype=np.int64)
    xp[:, :, ph:ph + H, pw:pw + W] = xs
    for i in range(KH):
        for j in range(3):
            patch = xp[:, :, i * dh:i * dh + (OH - 1) * sh + 1:sh, j:j + (OW - 1) * sw + 1:sw]
            out += np.einsum("bchw,nc->bnhw", patch, w[:, :, i, j])
    return out


CASES = [
    # B, cin, H, W, OC, KH, sh, ph, pw, dh, S
    (2, 8, 6, 8, 8, 3, 1, 1, 1, 1, 1),     # 3 x 3 "same"
    (1, 16, 7, 6, 20, 3, 1, 0, 0, 1, 1),    # "valid": OW = 4
    (1, 24, 9, 10, 8, 3, 1, 1, 1, 1, 2),    # 72 rows: ragged last tile, two splits
    (1, 8, 8, 4, 8, 1, 1, 0, 1, 1, 1),      # 1 x 3, W = 4 (the window IS the row)
    (1, 8, 9, 12, 8, 3, 2, 1, 2, 1, 1),     # stride 2 along the height, two columns of padding: OW = 14
    (1, 8, 10, 8, 8, 3, 1, 2, 1, 2, 1),     # dilation 2 along the height
    (3, 8, 5, 6, 136, 5, 1, 2, 1, 1, 1),    # 5 x 3, two channel tiles, M = 90 (tail)
]
# one pixel per thread: (B, cin, H, W, OC, KH, sh, ph, pw, dh, S, sw)
SINGLE_CASES = [
    (2, 8, 9, 9, 8, 3, 2, 1, 1, 1, 1, 2),     # the downsampling 3 x 3: stride 2 both ways, OW = 5
    (1, 16, 7, 7, 12, 3, 1, 1, 1, 1, 1, 1),   # stride 1 with an odd OW = 7 (the 7 x 7 maps of a ResNet's last stage)
    (1, 24, 6, 11, 8, 3, 1, 0, 2, 1, 2, 3),   # stride 3, two columns of padding, two splits, ragged last tile
    (1, 8, 5, 4, 8, 1, 1, 0, 0, 1, 1, 1),     # 1 x 3 "valid" on W = 4: OW = 2 - forced onto the single form below
    (2, 8, 6, 8, 8, 3, 1, 1, 1, 1, 1, 1),     # a pair-eligible geometry on the single form (QUANTO_HIP_CONV_ROWS=3)
]


def lds_bank_model():
    def ways(addrs, ndw, mod):
        c = {}
        for a in addrs:
            for i in range(ndw):
                c.setdefault(((a // 4) + i) % mod, set()).add(a)
        return max(len(v) for v in c.values())

    worst_r = 0
    for base, frag, kk, grp in itertools.product((0, 64), range(4), range(3), GROUPS_B128):
        worst_r = max(worst_r, ways([(base + frag * 16 + (l & 15)) * RS + (((l >> 4) ^ ((l >> 2) & 3)) << 4) + kk * 64 for l in grp], 4, 64))
    worst_a = 0
    for wave, px, j, g in itertools.product(range(8), range(2), range(3), range(4)):
        worst_a = max(worst_a, ways([(2 * l) * RS + (((wave >> 1) ^ ((l >> 1) & 3)) << 4) + (wave & 1) * 8 + px * RS + 64 * j for l in range(16 * g, 16 * g + 16)], 2, 32))
    worst_b = 0
    for wave, j, g in itertools.product(range(8), range(3), range(8)):
        tids = [wave * 64 + l for l in range(8 * g, 8 * g + 8)]
        worst_b = max(worst_b, ways([(t >> 2) * RS + (((t & 3) ^ ((t >> 4) & 3)) << 4) + 64 * j for t in tids], 4, 32))
    return worst_r, worst_a, worst_b


def lds_bank_model_single():
    """ds_write_b128 of the one-pixel form: eight contiguous lanes per LDS cycle, 32 banks"""
    worst = 0
    for wave, j, g in itertools.product(range(8), range(3), range(8)):
        tids = [wave * 64 + l for l in range(8 * g, 8 * g + 8)]
        addrs = [(t & 127) * RS + (((t >> 7) ^ ((t >> 2) & 3)) << 4) + 64 * j for t in tids]
        c = {}
        for a in addrs:
            for i in range(4):
                c.setdefault(((a // 4) + i) % 32, set()).add(a)
        worst = max(worst, max(len(v) for v in c.values()))
    return worst


def run_case(case, seed=0, single=None):
    B, cin, H, W, OC, KH, sh, ph, pw, dh, S = case[:11]
    sw = case[11] if len(case) > 11 else 1
    OH = (H + 2 * ph - dh * (KH - 1) - 1) // sh + 1
    OW = (W + 2 * pw - 3) // sw + 1
    rng = np.random.default_rng(seed)
    x16 = rng.integers(-300, 300, size=(B, cin, H, W), dtype=np.int16).view(np.uint16)
    wq = rng.integers(-128, 128, size=(OC, cin * KH * 3), dtype=np.int8)
    got = conv_rows_model(x16, wq, B, cin, H, W, OC, KH, OH, OW, sh, ph, pw, dh, S, sw, single)
    want = direct(x16, wq, B, cin, H, W, OC, KH, OH, OW, sh, ph, pw, dh, sw)
    return np.array_equal(got, want)


def main():
    print("LDS conflict degree: fragment ds_read_b128 %d, pixel ds_write_b64 %d, weight ds_write_b128 %d" % lds_bank_model(), "; one-pixel form ds_write_b128", lds_bank_model_single())
    for case in CASES:
        print(case, "ok" if run_case(case) else "MISMATCH")
    for case in SINGLE_CASES:
        print("one pixel per thread", case, "ok" if run_case(case, single=True) else "MISMATCH")


if __name__ == "__main__":
    main()
