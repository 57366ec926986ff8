#include "baseot.h"

#include <openssl/bn.h>
#include <openssl/ec.h>
#include <openssl/obj_mac.h>
#include <openssl/rand.h>
#include <openssl/sha.h>
#include <string.h>

#define PT_LEN 33   /* compressed P-256 point */

static void kdf(const EC_GROUP *g, const EC_POINT *p, int j, uint8_t out[16], BN_CTX *ctx) {
    uint8_t buf[4 + PT_LEN], h[32];
    buf[0] = (uint8_t)j; buf[1] = (uint8_t)(j >> 8); buf[2] = 0x4f; buf[3] = 0x54;
    EC_POINT_point2oct(g, p, POINT_CONVERSION_COMPRESSED, buf + 4, PT_LEN, ctx);
    SHA256(buf, sizeof buf, h);
    memcpy(out, h, 16);
}

int baseot_ext_receiver(node *n, int peer, uint8_t seeds0[128][16], uint8_t seeds1[128][16]) {
    int rc = 1;
    BN_CTX *ctx = BN_CTX_new();
    EC_GROUP *g = EC_GROUP_new_by_curve_name(NID_X9_62_prime256v1);
    BIGNUM *order = BN_new(), *a = BN_new();
    EC_POINT *A = EC_POINT_new(g), *B = EC_POINT_new(g), *T = EC_POINT_new(g), *negA = EC_POINT_new(g);
    uint8_t buf[128 * PT_LEN];
    EC_GROUP_get_order(g, order, ctx);
    BN_rand_range(a, order);
    EC_POINT_mul(g, A, a, 0, 0, ctx);                     /* A = aG */
    EC_POINT_point2oct(g, A, POINT_CONVERSION_COMPRESSED, buf, PT_LEN, ctx);
    if (net_send(n, peer, buf, PT_LEN)) goto done;
    if (net_recv(n, peer, buf, sizeof buf)) goto done;
    EC_POINT_copy(negA, A);
    EC_POINT_invert(g, negA, ctx);
    for (int j = 0; j < 128; j++) {
        if (!EC_POINT_oct2point(g, B, buf + j * PT_LEN, PT_LEN, ctx)) goto done;
        EC_POINT_mul(g, T, 0, B, a, ctx);                 /* a * B_j */
        kdf(g, T, j, seeds0[j], ctx);
        EC_POINT_add(g, T, B, negA, ctx);                 /* B_j - A */
        EC_POINT_mul(g, T, 0, T, a, ctx);
        kdf(g, T, j, seeds1[j], ctx);
    }
    rc = 0;
done:
    EC_POINT_free(A); EC_POINT_free(B); EC_POINT_free(T); EC_POINT_free(negA);
    BN_free(order); BN_clear_free(a); EC_GROUP_free(g); BN_CTX_free(ctx);
    return rc;
}

int baseot_ext_sender(node *n, int peer, uint8_t delta[16], uint8_t seeds[128][16]) {
    int rc = 1;
    BN_CTX *ctx = BN_CTX_new();
    EC_GROUP *g = EC_GROUP_new_by_curve_name(NID_X9_62_prime256v1);
    BIGNUM *order = BN_new(), *b = BN_new();
    EC_POINT *A = EC_POINT_new(g), *B = EC_POINT_new(g), *T = EC_POINT_new(g);
    uint8_t buf[128 * PT_LEN];
    EC_GROUP_get_order(g, order, ctx);
    if (RAND_bytes(delta, 16) != 1) goto done;
    if (net_recv(n, peer, buf, PT_LEN)) goto done;
    if (!EC_POINT_oct2point(g, A, buf, PT_LEN, ctx)) goto done;
    for (int j = 0; j < 128; j++) {
        int c = (delta[j >> 3] >> (j & 7)) & 1;
        BN_rand_range(b, order);
        EC_POINT_mul(g, B, b, 0, 0, ctx);                 /* b_j G */
        if (c) EC_POINT_add(g, B, B, A, ctx);             /* + c_j A */
        EC_POINT_point2oct(g, B, POINT_CONVERSION_COMPRESSED, buf + j * PT_LEN, PT_LEN, ctx);
        EC_POINT_mul(g, T, 0, A, b, ctx);                 /* b_j A */
        kdf(g, T, j, seeds[j], ctx);
    }
    if (net_send(n, peer, buf, sizeof buf)) goto done;
    rc = 0;
done:
    EC_POINT_free(A); EC_POINT_free(B); EC_POINT_free(T);
    BN_free(order); BN_clear_free(b); EC_GROUP_free(g); BN_CTX_free(ctx);
    return rc;
}
