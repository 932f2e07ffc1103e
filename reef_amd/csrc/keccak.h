// SHAKE256 (FIPS 202) on the host: the extendable-output stream of key derivation (row N1) is sequential by
// definition -- every 136-byte block is a Keccak-f[1600] permutation of the previous state -- so it is produced by the
// calling thread (4 MiB for 2^17 generators, a few milliseconds) and uploaded; the 2n maps to the curve run on the GPU.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string.h>

namespace reef {

inline void keccak_f1600(uint64_t st[25]) {
    static const uint64_t RC[24] = {0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL,
                                    0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL,
                                    0x0000000080008009ULL, 0x000000008000000aULL, 0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL,
                                    0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
                                    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
    static const int ROT[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
    static const int PIL[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};
    for (int round = 0; round < 24; ++round) {
        uint64_t bc[5];
        for (int i = 0; i < 5; ++i) bc[i] = st[i] ^ st[i + 5] ^ st[i + 10] ^ st[i + 15] ^ st[i + 20];
        for (int i = 0; i < 5; ++i) {
            const uint64_t t = bc[(i + 4) % 5] ^ ((bc[(i + 1) % 5] << 1) | (bc[(i + 1) % 5] >> 63));
            for (int j = 0; j < 25; j += 5) st[j + i] ^= t;
        }
        uint64_t t = st[1];
        for (int i = 0; i < 24; ++i) {
            const int j = PIL[i];
            const uint64_t b = st[j];
            st[j] = (t << ROT[i]) | (t >> (64 - ROT[i]));
            t = b;
        }
        for (int j = 0; j < 25; j += 5) {
            for (int i = 0; i < 5; ++i) bc[i] = st[j + i];
            for (int i = 0; i < 5; ++i) st[j + i] ^= (~bc[(i + 1) % 5]) & bc[(i + 2) % 5];
        }
        st[0] ^= RC[round];
    }
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
