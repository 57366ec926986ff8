/* pmsg.h -- the TI-mode wire message of the reference (src/protobuf/ *.proto, src/phase1.c:100-145):
 *     message msg { repeated uintW vector = 1 [packed = true]; required uintW value = 2; }
 * framed as an 8-byte host-endian size_t length followed by the protobuf bytes.
 * Hand-written proto2 encoder/decoder (protobuf-c is not available here); W = 32 or 64 only
 * changes the declared type, the varint wire encoding is the same. */
#ifndef LINREG_PMSG_H
#define LINREG_PMSG_H
#include <stddef.h>
#include <stdint.h>
size_t pmsg_packed_size(const uint64_t *vector, size_t n, uint64_t value);
size_t pmsg_pack(const uint64_t *vector, size_t n, uint64_t value, uint8_t *out);
/* returns 0 on success; *vector is malloc'd (caller frees) */
int pmsg_unpack(const uint8_t *buf, size_t len, uint64_t **vector, size_t *n, uint64_t *value);
/* into a caller-owned vector of cap words; more elements than cap is an error */
int pmsg_unpack_into(const uint8_t *buf, size_t len, uint64_t *vector, size_t cap, size_t *n, uint64_t *value);
#endif
