// mock_sdb.h -- knobs and the answer function of the CPU stand-in for libsemadb_amd (mock_sdb.cpp); test infrastructure
#pragma once
#include <cstdint>

// a search whose searchSize is this value fails as a device failure would (the whole batch: SDB_ERR_DEVICE)
#define MOCK_FAILING_SEARCH_SIZE 66u

// what the mock answers to one query (ids[limit], dists[limit]); filter: the query's ascending id list or NULL
void mock_expect(const float *q, uint32_t dim, uint32_t limit, const uint64_t *filter, uint64_t n_filter, uint64_t *ids,
                 float *dists, uint32_t *count);
void mock_set_latency_us(uint32_t lo, uint32_t hi);  // how long a device call sleeps (uniform in [lo, hi])
void mock_fail_host_alloc(int on);                   // sdb_host_alloc fails: the hosts fall back to pageable memory
uint64_t mock_violations(void);                      // breaches of the calling contract the mock has seen
const char *mock_first_violation(void);
uint64_t mock_search_calls(void);
