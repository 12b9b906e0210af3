// pg_rank_image.h -- GROUP BY on a raw column whose values do not fit the int dictId domain: FLOAT / DOUBLE columns, INT / LONG columns
// spanning more than 31 bits.  NoDictionarySingleColumnGroupKeyGenerator keys such a column by VALUE in a hash map
// (core/query/aggregation/groupby/NoDictionarySingleColumnGroupKeyGenerator.java:100-135, :240-300: Int / Long / Float / Double2IntOpenHashMap;
// NoDictionaryMultiColumnGroupKeyGenerator for tuples).  On the device the column gets a DICTIONARY built from its own values the first
// time it is grouped by -- the distinct values in ascending order (one radix sort + unique over order-preserving 64-bit images of the
// values) -- and a RANK IMAGE: the fixed-bit stream of every doc's rank in that dictionary.  From then on the column is an ordinary
// dictionary-encoded key for every group-by kernel (LDS table, partitioned, hashed holders): nothing downstream knows the difference.
//   image of a value: INT / LONG  v ^ 2^63;  FLOAT (widened exactly) / DOUBLE  bits >= 0 ? bits | 2^63 : ~bits, every NaN the canonical one
//   (Double.compare's order: -0.0 < 0.0, NaN above +Infinity; fastutil's double maps key by doubleToLongBits: one NaN, two zeros).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "../../include/pinot_gpu.h"

namespace pg {

// vkind: ValueKind of the raw column (pg_device.h).  The values area `d_raw` holds big-endian 4- or 8-byte values, doc order.
// Out: the dictionary on the device (order images, ascending; the caller owns it: hipFree), its host copy, the bit-packed rank image
// (lane-private tile layout of every fixed-bit column, padded to whole tiles + 64 bytes; the caller owns it) and its width / cardinality.
// PG_ERR_UNSUPPORTED when the column has 2^31 - 1 or more distinct values.
pg_status build_rank_image(const uint8_t* d_raw, int vkind, long long num_docs, int num_tiles, int num_cus, unsigned long long** out_d_dict,
                           std::vector<unsigned long long>* out_h_dict, uint8_t** out_image, size_t* out_image_bytes, int* out_bits, int* out_cardinality,
                           const char** out_error);

// the raw 64-bit pattern behind an order image: the long value (INT / LONG), the IEEE bits of the double (FLOAT widened / DOUBLE)
inline long long rank_image_value_bits(unsigned long long key, bool floating) {
  if (!floating) return (long long)(key ^ (1ull << 63));
  return (long long)((key >> 63) ? (key & ~(1ull << 63)) : ~key);
}

}  // namespace pg
