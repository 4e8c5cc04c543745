//go:build mi355x

// Package distance: the reference's own override mechanism (distance_amd64.go:19-27 reassigns the package-level
// func vars in an arch-gated init()) pointed at the MI355X.
package distance

/*
#cgo CFLAGS: -I${SRCDIR}/../third_party/semadb_amd/include
#cgo LDFLAGS: -lsemadb_amd
#include "semadb_amd.h"
*/
import "C"

import "unsafe"

func init() {
	var n C.int
	if C.sdb_device_count(&n) != C.SDB_OK || n <= 0 {
		return // no MI355X: keep whatever distance_amd64.go installed
	}
	// dotProductImpl returns the plain dot product; dotProductDistance / cosineDistance wrap it (distance.go:19-25)
	dotProductImpl = func(x, y []float32) float32 { return -gpuDist(C.SDB_METRIC_DOT, x, y) }
	euclideanDistance = func(x, y []float32) float32 { return gpuDist(C.SDB_METRIC_EUCLIDEAN, x, y) }
}

// One pair per launch: only there for interface completeness -- k-means and the flat index call FloatDistFunc
// pair by pair.  The search and insert paths never come through here; they use the batched entry points.
func gpuDist(metric C.int, x, y []float32) float32 {
	if len(x) == 0 {
		return 0
	}
	var out C.float
	C.sdb_distance_batch(metric, C.uint32_t(len(x)), (*C.float)(unsafe.Pointer(&x[0])), 1,
		(*C.float)(unsafe.Pointer(&y[0])), 1, &out, C.SDB_MEM_HOST, 0, nil)
	return float32(out)
}
