/*
 * rs_oracle.c -- CPU restatement of rs-mrzip's encoder: CCSDS RS(255,223) parity in
 * Berlekamp's dual basis + burst interleave + BLAKE2b trailer.
 * TEST INFRASTRUCTURE ONLY (see mrz_oracle.h).  Parity: pinned against
 * oracle/_ref/librs_ref.so (the reference's own rs-mrzip/reed-solomon.c compiled in
 * place) in tests/test_oracle_golden.py.
 *
 * Restates rs-mrzip/reed-solomon.c:115-141 (rse32), :311-321 (scatter) and
 * rs-mrzip/rs-mrzip.c:119-158 (encode).  Tables are generated, not copied:
 * GF(256) over p(x) = x^8+x^7+x^2+x+1 with alpha = 2 (reed-solomon.c:23-57), the CCSDS
 * generator g(x) = prod_{j=112..143} (x - alpha^(11 j)) (the Gg[] of :60-61 in index
 * form), and the dual-basis maps from the 8 basis images (taltab/tal1tab, :86-113).
 */
#include <stdlib.h>
#include <string.h>

#include "mrz_oracle.h"

#define RS_ROWS 8176 /* BLK_LEN = 16 * 511, rs-mrzip/reed-solomon.h:31 */
#define RS_K 223
#define RS_N 255

static uint8_t rs_exp[256], rs_log[256], rs_tal[256], rs_tal1[256], rs_gen[33];
static int rs_ready;

static uint8_t gf_mul(uint8_t a, uint8_t b) {
    if (!a || !b) return 0;
    return rs_exp[(rs_log[a] + rs_log[b]) % 255];
}

static void rs_init(void) {
    unsigned v = 1;
    for (int i = 0; i < 255; i++) {
        rs_exp[i] = (uint8_t)v;
        rs_log[v] = (uint8_t)i;
        v <<= 1;
        if (v & 0x100) v ^= 0x187;
    }
    rs_exp[255] = 0;
    rs_log[0] = 255;
    /* g(x) = prod (x - alpha^(11 j)), j = 112..143, coefficients in polynomial form */
    uint8_t g[33] = { 1 };
    for (int j = 112, deg = 0; j <= 143; j++, deg++) {
        const uint8_t root = rs_exp[(11 * j) % 255];
        g[deg + 1] = 0;
        for (int k = deg + 1; k > 0; k--) g[k] = g[k - 1] ^ gf_mul(g[k], root);
        g[0] = gf_mul(g[0], root);
    }
    memcpy(rs_gen, g, 33);
    /* conventional -> dual basis: linear map given by the images of the 8 basis bits */
    static const uint8_t basis[8] = { 0x8d, 0xef, 0xec, 0x86, 0xfa, 0x99, 0xaf, 0x7b };
    for (int i = 0; i < 256; i++) {
        uint8_t t = 0;
        for (int k = 0; k < 8; k++)
            if (i & (1 << k)) t ^= basis[7 - k];
        rs_tal[i] = t;
    }
    for (int i = 0; i < 256; i++) rs_tal1[rs_tal[i]] = (uint8_t)i;
    rs_ready = 1;
}

/* parity of one 223-byte row, dual basis in and out (rse32, reed-solomon.c:115-141) */
void mrzo_rs_parity(const uint8_t data[223], uint8_t parity[32]) {
    if (!rs_ready) rs_init();
    uint8_t bb[32] = { 0 };
    for (int i = RS_K - 1; i >= 0; i--) {
        const uint8_t fb = rs_tal1[data[i]] ^ bb[31];
        for (int j = 31; j > 0; j--) bb[j] = bb[j - 1] ^ gf_mul(rs_gen[j], fb);
        bb[0] = gf_mul(rs_gen[0], fb);
    }
    for (int j = 0; j < 32; j++) parity[j] = rs_tal[bb[j]];
}

/* whole stream: what `rs-mrzip` writes for n input bytes (rs-mrzip.c:119-158).
 * out must hold mrzo_rs_encoded_size(n) bytes. */
int64_t mrzo_rs_encoded_size(int64_t n) {
    const int64_t burst_in = (int64_t)RS_K * RS_ROWS;
    return (n / burst_in + 1) * (int64_t)RS_N * RS_ROWS + 64 + 4; /* feof() is only set by a short read */
}

int mrzo_rs_encode(const uint8_t *in, int64_t n, uint8_t *out) {
    if (!rs_ready) rs_init();
    const int64_t burst_in = (int64_t)RS_K * RS_ROWS, burst_out = (int64_t)RS_N * RS_ROWS;
    const int64_t nbursts = n / burst_in + 1;
    mrzo_blake2b h;
    mrzo_blake2b_init(&h, 64);
    unsigned k_i = 0xFFFF, k_j = 0xFFFF;
    uint8_t row[RS_N];
    int64_t pos = 0;
    for (int64_t b = 0; b < nbursts; b++) {
        uint8_t *dst = out + b * burst_out;
        for (int r = 0; r < RS_ROWS; r++) {
            int64_t got = n - pos;
            if (got > RS_K) got = RS_K;
            if (got < 0) got = 0;
            memcpy(row, in + pos, (size_t)got);
            pos += got;
            if (got < RS_K) {
                memset(row + got, 0, (size_t)(RS_K - got));
                if (k_i == 0xFFFF && k_j == 0xFFFF) {
                    k_i = (unsigned)r;
                    k_j = (unsigned)got;
                }
            }
            mrzo_blake2b_update(&h, row, RS_K);
            mrzo_rs_parity(row, row + RS_K);
            /* scatter (reed-solomon.c:311-321): column c of row r lands at c * BLK_LEN + r */
            for (int c = 0; c < RS_N; c++) dst[(int64_t)c * RS_ROWS + r] = row[c];
        }
    }
    uint8_t *tail = out + nbursts * burst_out;
    mrzo_blake2b_final(&h, tail);
    tail[64] = (uint8_t)(k_i & 0xFF);
    tail[65] = (uint8_t)(k_i >> 8);
    tail[66] = (uint8_t)(k_j & 0xFF);
    tail[67] = (uint8_t)(k_j >> 8);
    return 0;
}

/* table access for tests */
void mrzo_rs_tables(uint8_t exp_[256], uint8_t log_[256], uint8_t tal[256], uint8_t tal1[256], uint8_t gen_index[33]) {
    if (!rs_ready) rs_init();
    memcpy(exp_, rs_exp, 256);
    memcpy(log_, rs_log, 256);
    memcpy(tal, rs_tal, 256);
    memcpy(tal1, rs_tal1, 256);
    for (int j = 0; j < 33; j++) gen_index[j] = rs_log[rs_gen[j]];
}
