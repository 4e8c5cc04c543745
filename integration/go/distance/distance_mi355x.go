//go:build mi355x

// Package distance (addition).  The per-pair functions stay what distance_amd64.go:19-27 installs: the AVX2 assembly
// (asm.Dot, asm.SquaredEuclideanDistance).  They are bit-identical to the MI355X kernels -- the kernels were written to
// reproduce exactly that arithmetic -- and a pair at a time a host SIMD routine is four orders of magnitude cheaper
// than a kernel launch, so nothing that calls a FloatDistFunc pair by pair (k-means of other quantizers, haversine
// neighbours, the package's own tests) is rerouted.  The GPU is offered only in the shape it is good at: a whole
// query x candidate block per call.  The search and insert paths do not come through here at all; they use the index
// entry points (shard/index/vamana).
package distance

/*
#cgo CFLAGS: -I${SRCDIR}/../third_party/semadb_amd/include
#cgo LDFLAGS: -lsemadb_amd
#include "semadb_amd.h"
*/
import "C"

import (
	"fmt"
	"unsafe"
)

var metricCodes = map[string]C.int{"euclidean": C.SDB_METRIC_EUCLIDEAN, "cosine": C.SDB_METRIC_COSINE, "dot": C.SDB_METRIC_DOT}

// BatchDistance returns out[q*nc+c] = GetFloatDistanceFn(name)(queries[q], candidates[c]) for row-major blocks of
// dim-float vectors, computed on device `device` with the reference's summation order (same bits as the assembly).
func BatchDistance(name string, dim int, queries, candidates []float32, device int) ([]float32, error) {
	code, ok := metricCodes[name]
	if !ok {
		return nil, fmt.Errorf("unknown float32 distance function: %s", name) // distance.go:81
	}
	if dim <= 0 || len(queries)%dim != 0 || len(candidates)%dim != 0 {
		return nil, fmt.Errorf("blocks of %d and %d floats are not whole rows of %d", len(queries), len(candidates), dim)
	}
	nq, nc := len(queries)/dim, len(candidates)/dim
	out := make([]float32, nq*nc)
	if nq == 0 || nc == 0 {
		return out, nil
	}
	rc := C.sdb_distance_batch(code, C.uint32_t(dim), (*C.float)(unsafe.Pointer(&queries[0])), C.uint64_t(nq),
		(*C.float)(unsafe.Pointer(&candidates[0])), C.uint64_t(nc), (*C.float)(unsafe.Pointer(&out[0])), C.SDB_MEM_HOST,
		C.int(device), nil)
	if rc != C.SDB_OK {
		return nil, fmt.Errorf("batch distance: %s (status %d)", C.GoString(C.sdb_last_error()), int(rc))
	}
	return out, nil
}
