/* baseot.h -- the kappa = 128 base OTs that seed the IKNP extension (Obliv-C does this with
 * Naor-Pinkas inside honestOTExt{Sender,Recver}New, e.g. src/input.c:28,67).  Here: the
 * Chou-Orlandi "simplest OT" over NIST P-256 with OpenSSL, SHA-256 as the key-derivation hash.
 * Host-side public-key work; the extension itself runs on the GPU (liblinreg_gc: lgc_ot_*). */
#ifndef LINREG_BASEOT_H
#define LINREG_BASEOT_H
#include <stdint.h>
#include "net.h"
/* the party that will be the extension RECEIVER (base-OT sender): gets both seeds per column */
int baseot_ext_receiver(node *n, int peer, uint8_t seeds0[128][16], uint8_t seeds1[128][16]);
/* the party that will be the extension SENDER (base-OT receiver): random delta, one seed per column */
int baseot_ext_sender(node *n, int peer, uint8_t delta[16], uint8_t seeds[128][16]);
#endif
