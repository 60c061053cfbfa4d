/* ecgpu_status.h -- status codes of the ecgpu C ABI (part of include/ecgpu.h; a header of its own so that the
 * device-side translation units depend on the codes only, not on every prototype of the ABI). */
#ifndef ECGPU_STATUS_H
#define ECGPU_STATUS_H

/* BLST_ERROR numbering (blst bindings; reference crypto/bls.rs:48-62) */
#define ECGPU_SUCCESS 0
#define ECGPU_BAD_ENCODING 1
#define ECGPU_POINT_NOT_ON_CURVE 2
#define ECGPU_POINT_NOT_IN_GROUP 3
#define ECGPU_AGGR_TYPE_MISMATCH 4
#define ECGPU_VERIFY_FAIL 5
#define ECGPU_PK_IS_INFINITY 6
#define ECGPU_BAD_SCALAR 7
/* Error identity (crypto/bls.rs:69-76,100-111,119-131).  The reference raises `Error::BLST(..)` for a BLST_ERROR met
 * while CONVERTING a key or a signature (`TryFrom<&PublicKey>` = key_validate, `TryFrom<&Signature>` = from_bytes) and
 * collapses every non-SUCCESS result of blst's own verify call to `Error::InvalidSignature`.  Two BLST_ERROR values can
 * come from either place, so the verify-side ones carry ECGPU_IN_VERIFY:
 *   3 POINT_NOT_IN_GROUP  a public key outside G1 (conversion)          -> Error::BLST("point not in group")
 *   0x43                  the signature outside G2, found by verify's group check -> Error::InvalidSignature
 *   6 PK_IS_INFINITY      a public key that decodes to infinity (conversion) -> Error::BLST("public key is infinity")
 *   0x46                  the keys sum to infinity inside fast_aggregate_verify -> Error::InvalidSignature
 * Rule for a binding: 0 -> Ok; 1, 2, 3, 6 -> Error::BLST(BLSTError(code)); any other positive value ->
 * Error::InvalidSignature (4 AGGR_TYPE_MISMATCH and 5 VERIFY_FAIL only ever come from the verify call). */
#define ECGPU_IN_VERIFY 0x40
#define ECGPU_VERIFY_POINT_NOT_IN_GROUP (ECGPU_IN_VERIFY | ECGPU_POINT_NOT_IN_GROUP)
#define ECGPU_VERIFY_PK_IS_INFINITY (ECGPU_IN_VERIFY | ECGPU_PK_IS_INFINITY)
/* wrapper-level and backend conditions */
#define ECGPU_EMPTY_AGGREGATE (-100) /* Error::EmptyAggregate, crypto/bls.rs:80-82,136-138 */
#define ECGPU_ERR_NO_DEVICE (-1)
#define ECGPU_ERR_HIP (-2)
#define ECGPU_ERR_BAD_ARG (-3)
#define ECGPU_ERR_OOM (-4)

#endif /* ECGPU_STATUS_H */
