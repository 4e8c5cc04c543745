// stubs.hip -- entry points not implemented yet (replaced as the kernels land).
#include "common.h"
using namespace sdb;
extern "C" {
int sdb_kmeans_fit(float *, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, int, float *,
                   uint8_t *, uint32_t *, int, int, void *) {
  return fail(SDB_ERR_STATE, "kmeans_fit: not implemented yet");
}
int sdb_pq_create(uint32_t, uint32_t, uint32_t, uint32_t, int, sdb_pq **) { return fail(SDB_ERR_STATE, "pq: not implemented yet"); }
int sdb_pq_destroy(sdb_pq *) { return SDB_OK; }
int sdb_pq_fit(sdb_pq *, float *, uint32_t, const uint32_t *, int, uint8_t *, int, void *) { return fail(SDB_ERR_STATE, "pq: not implemented yet"); }
int sdb_pq_set_codebook(sdb_pq *, const float *, int) { return fail(SDB_ERR_STATE, "pq: not implemented yet"); }
int sdb_pq_get_codebook(const sdb_pq *, float *, float *) { return fail(SDB_ERR_STATE, "pq: not implemented yet"); }
int sdb_pq_encode(const sdb_pq *, const float *, uint64_t, uint8_t *, int, void *) { return fail(SDB_ERR_STATE, "pq: not implemented yet"); }
int sdb_pq_lut_distance(const sdb_pq *, const float *, uint64_t, const uint8_t *, uint64_t, float *, int, void *) { return fail(SDB_ERR_STATE, "pq: not implemented yet"); }
int sdb_pq_sym_distance(const sdb_pq *, const uint8_t *, const uint8_t *, uint64_t, float *, int, void *) { return fail(SDB_ERR_STATE, "pq: not implemented yet"); }
int sdb_index_attach_pq(sdb_index *, const sdb_pq *, void *) { return fail(SDB_ERR_STATE, "pq: not implemented yet"); }
}
