// SHAKE256 (FIPS 202) on the host: the extendable-output stream of key derivation (row N1) is sequential by
// definition -- every 136-byte block is a Keccak-f[1600] permutation of the previous state -- so it is produced by the
// calling thread (4 MiB for 2^17 generators, a few milliseconds) and uploaded; the 2n maps to the curve run on the GPU.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string.h>

namespace reef {

static inline uint64_t rotl64(uint64_t v, int r) { return (v << r) | (v >> (64 - r)); }
}  // namespace reef
#include "keccak_round.inc"
namespace reef {

// Keccak-f[1600] with the state in 25 lane variables, two rounds per loop iteration (a -> e -> a): the sponge is one serial
// chain, so what counts is the latency of a permutation -- 1.5x less than the table-driven loop this replaces (st[] in
// memory, pi as a 24-step move chain).  The round is generated from the FIPS 202 step mappings (tools/gen_keccak_round.py).
inline void keccak_f1600(uint64_t st[25]) {
    static const uint64_t RC[24] = {0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL,
                                    0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL,
                                    0x0000000080008009ULL, 0x000000008000000aULL, 0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL,
                                    0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
                                    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
    uint64_t a00 = st[0], a10 = st[1], a20 = st[2], a30 = st[3], a40 = st[4], a01 = st[5], a11 = st[6], a21 = st[7], a31 = st[8], a41 = st[9],
             a02 = st[10], a12 = st[11], a22 = st[12], a32 = st[13], a42 = st[14], a03 = st[15], a13 = st[16], a23 = st[17], a33 = st[18], a43 = st[19],
             a04 = st[20], a14 = st[21], a24 = st[22], a34 = st[23], a44 = st[24];
    uint64_t e00, e10, e20, e30, e40, e01, e11, e21, e31, e41, e02, e12, e22, e32, e42, e03, e13, e23, e33, e43, e04, e14, e24, e34, e44;
    uint64_t b00, b10, b20, b30, b40, b01, b11, b21, b31, b41, b02, b12, b22, b32, b42, b03, b13, b23, b33, b43, b04, b14, b24, b34, b44;
    uint64_t c0, c1, c2, c3, c4, d0, d1, d2, d3, d4;
    for (int round = 0; round < 24; round += 2) {
        REEF_KECCAK_ROUND(a, e, RC[round])
        REEF_KECCAK_ROUND(e, a, RC[round + 1])
    }
    st[0] = a00; st[1] = a10; st[2] = a20; st[3] = a30; st[4] = a40; st[5] = a01; st[6] = a11; st[7] = a21; st[8] = a31; st[9] = a41;
    st[10] = a02; st[11] = a12; st[12] = a22; st[13] = a32; st[14] = a42; st[15] = a03; st[16] = a13; st[17] = a23; st[18] = a33; st[19] = a43;
    st[20] = a04; st[21] = a14; st[22] = a24; st[23] = a34; st[24] = a44;
}

// SHAKE256 as a stream: absorb the whole input once, then squeeze any number of bytes in pieces.
struct Shake256 {
    static constexpr size_t RATE = 136;
    uint64_t st[25];
    size_t pos = RATE;                                     // bytes of the current block already handed out
    Shake256(const uint8_t *in, size_t in_len) {
        memset(st, 0, sizeof st);
        uint8_t *sb = reinterpret_cast<uint8_t *>(st);     // little-endian host (x86-64 / the GPU box)
        while (in_len >= RATE) {
            for (size_t i = 0; i < RATE; ++i) sb[i] ^= in[i];
            keccak_f1600(st);
            in += RATE;
            in_len -= RATE;
        }
        for (size_t i = 0; i < in_len; ++i) sb[i] ^= in[i];
        sb[in_len] ^= 0x1F;                                // SHAKE domain bits + first padding bit
        sb[RATE - 1] ^= 0x80;
    }
    void squeeze(uint8_t *out, size_t out_len) {
        const uint8_t *sb = reinterpret_cast<const uint8_t *>(st);
        while (out_len) {
            if (pos == RATE) {
                keccak_f1600(st);
                pos = 0;
            }
            const size_t take = out_len < RATE - pos ? out_len : RATE - pos;
            memcpy(out, sb + pos, take);
            out += take;
            out_len -= take;
            pos += take;
        }
    }
};

// out[0..out_len) = SHAKE256(in[0..in_len))
inline void shake256(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len) {
    Shake256 x(in, in_len);
    x.squeeze(out, out_len);
}

}  // namespace reef
