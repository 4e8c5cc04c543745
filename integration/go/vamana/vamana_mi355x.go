//go:build mi355x

// Package vamana: drop-in replacement of shard/index/vamana with the index state in one MI355X's HBM.
// Same exported surface as the reference package (vamana.go:28,54,83,87,122-125,127,278; node.go:142).
package vamana

/*
#cgo CFLAGS: -I${SRCDIR}/../../../third_party/semadb_amd/include
#cgo LDFLAGS: -lsemadb_amd
#include <stdlib.h>
#include "semadb_amd.h"
*/
import "C"

import (
	"context"
	"fmt"
	"hash/fnv"
	"math"
	"math/rand/v2"
	"sync"
	"time"
	"unsafe"

	"github.com/RoaringBitmap/roaring/roaring64"
	"github.com/semafind/semadb/diskstore"
	"github.com/semafind/semadb/models"
)

// vamana.go:28
const STARTID = 1

// vamana.go:31
const MAXNODEIDKEY = "_vamanaMaxNodeId"

// vamana.go:122-125; Vector == nil means delete
type IndexVectorChange struct {
	Id     uint64
	Vector []float32
}

type IndexVamana struct {
	parameters models.IndexVectorVamanaParameters
	bucket     diskstore.Bucket
	h          *C.sdb_index
	pq         *C.sdb_pq // product quantizer of the vector store (vectorstore.New), nil for the plain store
	pqFitted   bool
	batcher    *searchBatcher // coalesces concurrent Search calls
	// writers exclude each other, like the shard's write transaction (shard/cache/manager.go:183-240).  Searches do
	// NOT take this lock: the library serves them from the last committed graph while a write is open.
	mu sync.Mutex
}

func lastErr(what string, rc C.int) error {
	return fmt.Errorf("%s: %s (status %d)", what, C.GoString(C.sdb_last_error()), int(rc))
}

var metricCode = map[string]C.uint32_t{
	models.DistanceEuclidean: C.SDB_METRIC_EUCLIDEAN,
	models.DistanceCosine:    C.SDB_METRIC_COSINE,
	models.DistanceDot:       C.SDB_METRIC_DOT,
}

// deviceForShard pins an index to one of the node's GPUs.  The shard manager passes the shard's directory
// name as the index name prefix (shard/shard.go:96-104); hashing it spreads shards over the devices and is
// stable across restarts.  A deployment that wants an explicit map sets SEMADB_MI355X_DEVICE_<name>.
func deviceForShard(name string) int {
	var n C.int
	if rc := C.sdb_device_count(&n); rc != C.SDB_OK || n <= 0 {
		return 0
	}
	h := fnv.New32a()
	h.Write([]byte(name))
	return int(h.Sum32() % uint32(n))
}

// setupStartNode's random unit vector (vamana.go:99-110)
func randomUnitVector(d int) []float32 {
	v := make([]float32, d)
	sum := float32(0)
	for i := range v {
		v[i] = rand.Float32()*2 - 1
		sum += v[i] * v[i]
	}
	norm := 1 / float32(math.Sqrt(float64(sum)))
	for i := range v {
		v[i] *= norm
	}
	return v
}

// NewIndexVamana (vamana.go:54-81): device index + vector store, filled from the bucket.
// TwoPrecisionSearch switches SDB_TUNE_SKETCH on for every index created afterwards (a server sets it once from its
// configuration): the device keeps a float16 copy of the rows (+ 50 % of their memory) and a batch search reads a
// neighbour's float32 row only when its float16 distance does not prove that AddWithLimit discards it
// (distset.go:184).  Same answers, bit for bit; 1.50 M against 1.07 M queries/s at 1M x 384.  Off like in the library.
var TwoPrecisionSearch = false

func NewIndexVamana(name string, params models.IndexVectorVamanaParameters, bucket diskstore.Bucket) (*IndexVamana, error) {
	mc, ok := metricCode[params.DistanceMetric]
	if !ok { // hamming / jaccard / haversine stay on the reference's CPU path
		return nil, fmt.Errorf("could not create vector store: distance %s is not on the MI355X path", params.DistanceMetric)
	}
	p := C.sdb_index_params{
		dim: C.uint32_t(params.VectorSize), metric: mc,
		search_size: C.uint32_t(params.SearchSize), degree_bound: C.uint32_t(params.DegreeBound),
		alpha: C.float(params.Alpha), device: C.int32_t(deviceForShard(name)), strict: 1,
	}
	v := &IndexVamana{parameters: params, bucket: bucket}
	if rc := C.sdb_index_create(&p, &v.h); rc != C.SDB_OK {
		return nil, lastErr("could not create device index", rc)
	}
	if q := params.Quantizer; q != nil && q.Type != models.QuantizerNone { // vectorstore.New (vectorstore.go:47-96)
		if q.Type != models.QuantizerProduct || q.Product == nil {
			C.sdb_index_destroy(v.h)
			return nil, fmt.Errorf("could not create vector store: quantizer %s is not on the MI355X path", q.Type)
		}
		if rc := C.sdb_pq_create(C.uint32_t(params.VectorSize), mc, C.uint32_t(q.Product.NumSubVectors),
			C.uint32_t(q.Product.NumCentroids), p.device, &v.pq); rc != C.SDB_OK {
			C.sdb_index_destroy(v.h)
			return nil, lastErr("could not create vector store", rc)
		}
	}
	if err := v.loadFromBucket(); err != nil {
		v.Close()
		return nil, fmt.Errorf("could not setup start node: %w", err)
	}
	if TwoPrecisionSearch { // (after the load: the copy is built from the rows that are there, commits keep it current)
		if rc := C.sdb_index_set_tuning(v.h, C.SDB_TUNE_SKETCH, 1); rc != C.SDB_OK {
			v.Close()
			return nil, lastErr("could not switch the two-precision search on", rc)
		}
	}
	v.batcher = newSearchBatcher(v, 1024, 200*time.Microsecond, 4) // batches in flight: see semadb_host.hpp (2 -> 4: 1.06 -> 1.18 M queries/s)
	return v, nil
}

// Close releases the HBM state (the reference relies on the GC; a device handle cannot).
func (v *IndexVamana) Close() {
	if v.batcher != nil {
		v.batcher.stop()
	}
	if v.h != nil {
		C.sdb_index_destroy(v.h)
		v.h = nil
	}
	if v.pq != nil {
		C.sdb_pq_destroy(v.pq)
		v.pq = nil
	}
}

// vamana.go:83-85
func (v *IndexVamana) SizeInMemory() int64 {
	var b C.int64_t
	C.sdb_index_size_in_memory(v.h, &b)
	return int64(b)
}

// vamana.go:87-91
func (v *IndexVamana) UpdateBucket(bucket diskstore.Bucket) { v.bucket = bucket }

// exists: vecStore.Exists (plain.go:21-24), a host-side table lookup in the library
func (v *IndexVamana) exists(ids []uint64) []bool {
	out := make([]bool, len(ids))
	if len(ids) == 0 {
		return out
	}
	flags := make([]C.uint8_t, len(ids))
	C.sdb_index_exists_batch(v.h, C.uint64_t(len(ids)), (*C.uint64_t)(unsafe.Pointer(&ids[0])), &flags[0])
	for i, f := range flags {
		out[i] = f != 0
	}
	return out
}

// getMany: vecStore.GetMany (plain.go:26-45) -- stored vectors of ids in request order, missing ids skipped
func (v *IndexVamana) getMany(ids []uint64) [][]float32 {
	if len(ids) == 0 {
		return nil
	}
	d := int(v.parameters.VectorSize)
	flat := make([]float32, len(ids)*d)
	found := make([]C.uint8_t, len(ids))
	if rc := C.sdb_index_get_vectors(v.h, C.uint64_t(len(ids)), (*C.uint64_t)(unsafe.Pointer(&ids[0])),
		(*C.float)(unsafe.Pointer(&flat[0])), &found[0]); rc != C.SDB_OK {
		return nil
	}
	out := make([][]float32, 0, len(ids))
	for i := range ids {
		if found[i] != 0 {
			out = append(out, flat[i*d:(i+1)*d])
		}
	}
	return out
}

// Search has the reference's signature (vamana.go:278).  Concurrent callers -- one goroutine per request,
// shard/cache/manager.go:163 -- are coalesced into device batches by the batcher.
func (v *IndexVamana) Search(ctx context.Context, q models.SearchVectorVamanaOptions, filter *roaring64.Bitmap) (*roaring64.Bitmap, []models.SearchResult, error) {
	if q.SearchSize < q.Limit { // search.go:23-25
		return nil, nil, fmt.Errorf("could not perform graph search: searchSize (%d) must be greater than k (%d)", q.SearchSize, q.Limit)
	}
	if len(q.Vector) != int(v.parameters.VectorSize) { // rejected upstream, models/search.go:198-200
		return nil, nil, fmt.Errorf("could not perform graph search: query vector length %d, index %d", len(q.Vector), v.parameters.VectorSize)
	}
	ids, dists, err := v.batcher.submit(ctx, q.Vector, q.Limit, q.SearchSize, filter)
	if err != nil {
		return nil, nil, fmt.Errorf("could not perform graph search: %w", err)
	}
	weight := float32(1) // vamana.go:289-292
	if q.Weight != nil {
		weight = *q.Weight
	}
	results := make([]models.SearchResult, 0, len(ids))
	set := roaring64.New()
	for i := range ids {
		d := dists[i]
		results = append(results, models.SearchResult{NodeId: ids[i], Distance: &d, HybridScore: -1 * d * weight}) // :300-304
		set.Add(ids[i])
	}
	return set, results, nil
}

// InsertUpdateDelete (vamana.go:127-263): same classification and order as :149-251 -- new ids are inserted
// first (one device call, rounds inside); the inbound edges of deleted AND updated ids are removed in one scan
// and the deleted nodes dropped; updated points are re-inserted one by one; Fit; flush.
func (v *IndexVamana) InsertUpdateDelete(ctx context.Context, points <-chan IndexVectorChange) <-chan error {
	errC := make(chan error, 1)
	go func() {
		errC <- v.insertUpdateDelete(ctx, points)
		close(errC)
	}()
	return errC
}

func (v *IndexVamana) insertUpdateDelete(ctx context.Context, points <-chan IndexVectorChange) error {
	v.mu.Lock()
	defer v.mu.Unlock()
	var changes []IndexVectorChange
	for p := range points {
		if p.Id == STARTID { // vamana.go:150-153
			return fmt.Errorf("cannot modify point with start id: %d", STARTID)
		}
		if p.Id == 0 { // :154-157
			return fmt.Errorf("invalid point id: %d", p.Id)
		}
		changes = append(changes, p)
	}
	if err := ctx.Err(); err != nil { // the reference checks the context between points (:199)
		return fmt.Errorf("context done while inserting: %w", err)
	}
	allIds := make([]uint64, len(changes))
	for i, c := range changes {
		allIds[i] = c.Id
	}
	stored := v.exists(allIds) // one lookup for the whole change list
	fresh := make(map[uint64]struct{})
	var insIds, updIds, delIds []uint64
	var insVecs, updVecs []float32
	for i, p := range changes {
		_, seen := fresh[p.Id]
		exists := stored[i] || seen
		switch {
		case !exists && p.Vector == nil: // nothing to do (:161-163)
		case !exists:
			insIds = append(insIds, p.Id)
			insVecs = append(insVecs, p.Vector...) // copied: the callee never retains Go memory either
			fresh[p.Id] = struct{}{}
		case p.Vector != nil: // update (:170-174)
			updIds = append(updIds, p.Id)
			updVecs = append(updVecs, p.Vector...)
		default: // delete (:175-179)
			delIds = append(delIds, p.Id)
		}
	}
	if len(insIds)+len(delIds)+len(updIds) == 0 {
		return nil
	}
	// one write transaction, like the shard's: concurrent searches see all of it or none of it
	if rc := C.sdb_index_begin_write(v.h); rc != C.SDB_OK {
		return lastErr("could not start the write", rc)
	}
	committed := false
	defer func() {
		// An error inside the transaction: roll it back.  The device keeps the committed copy of the graph that the
		// searches walk; abort_write restores the writer's copy from it, so the index is what it was before this call
		// and takes the next write.  (The cache manager scraps a shard after any error inside a write anyway,
		// manager.go:231-240; with the rollback it may just as well keep this one.)
		if !committed {
			C.sdb_index_abort_write(v.h)
		}
	}()
	if len(insIds) > 0 {
		if rc := C.sdb_index_insert_batch(v.h, C.uint64_t(len(insIds)), (*C.uint64_t)(unsafe.Pointer(&insIds[0])),
			(*C.float)(unsafe.Pointer(&insVecs[0])), C.SDB_MEM_HOST, 0, nil); rc != C.SDB_OK {
			return lastErr("could not distribute or insert points", rc)
		}
	}
	if gone := append(append([]uint64{}, delIds...), updIds...); len(gone) > 0 { // removeInboundEdges (:223-233)
		if rc := C.sdb_index_delete_batch(v.h, C.uint64_t(len(gone)), (*C.uint64_t)(unsafe.Pointer(&gone[0])), nil); rc != C.SDB_OK {
			return lastErr("could not remove inbound edges", rc)
		}
	}
	d := int(v.parameters.VectorSize)
	for i := range updIds { // :247-251 re-inserted sequentially
		if rc := C.sdb_index_insert_batch(v.h, 1, (*C.uint64_t)(unsafe.Pointer(&updIds[i])),
			(*C.float)(unsafe.Pointer(&updVecs[i*d])), C.SDB_MEM_HOST, 1, nil); rc != C.SDB_OK {
			return lastErr("could not re-insert updated point", rc)
		}
	}
	if rc := C.sdb_index_commit(v.h, nil); rc != C.SDB_OK {
		return lastErr("could not commit the write", rc)
	}
	committed = true
	if err := v.fit(); err != nil { // vecStore.Fit (:257-260)
		return fmt.Errorf("could not fit vector store: %w", err)
	}
	if err := v.flushToBucket(delIds); err != nil { // :265-276
		return err
	}
	// the reference frees deleted nodes at flush (node.go:129-134); here their rows stay behind as tombstones
	// until they are worth squeezing out
	var rows, dead C.uint64_t
	if C.sdb_index_row_usage(v.h, &rows, &dead) == C.SDB_OK && dead*4 > rows {
		if rc := C.sdb_index_compact(v.h); rc != C.SDB_OK {
			return lastErr("could not compact the index", rc)
		}
	}
	return nil
}

// EdgeScan (node.go:142-199), same signature: nodes with an edge into deleteSet, and valid nodes nobody
// points at.  The reference returns both in map order; so may this.
func (v *IndexVamana) EdgeScan(deleteSet map[uint64]struct{}) (toPrune, toSave []uint64, err error) {
	v.mu.Lock()
	defer v.mu.Unlock()
	del := make([]uint64, 0, len(deleteSet))
	for id := range deleteSet {
		del = append(del, id)
	}
	var nNodes, nEdges, maxId C.uint64_t
	C.sdb_index_stats(v.h, &nNodes, &nEdges, &maxId)
	toPrune = make([]uint64, int(nNodes)+1)
	toSave = make([]uint64, int(nNodes)+1)
	var np, ns C.uint64_t
	var dp *C.uint64_t
	if len(del) > 0 {
		dp = (*C.uint64_t)(unsafe.Pointer(&del[0]))
	}
	if rc := C.sdb_index_edge_scan(v.h, C.uint64_t(len(del)), dp, (*C.uint64_t)(unsafe.Pointer(&toPrune[0])),
		C.uint64_t(len(toPrune)), &np, (*C.uint64_t)(unsafe.Pointer(&toSave[0])), C.uint64_t(len(toSave)), &ns, nil); rc != C.SDB_OK {
		return nil, nil, lastErr("could not scan edges", rc)
	}
	return toPrune[:np], toSave[:ns], nil
}

// fit: productQuantizer.Fit (product.go:175-236) once the store holds TriggerThreshold points.
func (v *IndexVamana) fit() error {
	if v.pq == nil || v.pqFitted {
		return nil
	}
	pp := v.parameters.Quantizer.Product
	ids, vecs, _, _, err := v.exportVectors(true)
	if err != nil {
		return err
	}
	if len(ids) < pp.TriggerThreshold {
		return nil
	}
	M := pp.NumSubVectors
	first := make([]uint32, M) // kmeans.go:61-63: one random first centroid per sub-quantizer
	for i := range first {
		first[i] = uint32(rand.IntN(len(ids)))
	}
	codes := make([]uint8, len(ids)*M)
	// alias = 1: centroids are views into the data rows and the means overwrite them (kmeans.go:63,82,144);
	// vecs is this function's private copy, the slab keeps the stored vectors
	if rc := C.sdb_pq_fit(v.pq, (*C.float)(unsafe.Pointer(&vecs[0])), C.uint32_t(len(ids)),
		(*C.uint32_t)(unsafe.Pointer(&first[0])), 1, (*C.uint8_t)(unsafe.Pointer(&codes[0])), C.SDB_MEM_HOST, nil); rc != C.SDB_OK {
		return lastErr("kmeans", rc)
	}
	if rc := C.sdb_index_attach_pq(v.h, v.pq, nil); rc != C.SDB_OK { // every stored vector encoded (product.go:161-169) ...
		return lastErr("could not attach quantizer", rc)
	}
	if rc := C.sdb_index_set_codes(v.h, C.uint64_t(len(ids)), (*C.uint64_t)(unsafe.Pointer(&ids[0])),
		(*C.uint8_t)(unsafe.Pointer(&codes[0]))); rc != C.SDB_OK { // ... then the k-means labels (:216-218)
		return lastErr("could not store centroid ids", rc)
	}
	v.pqFitted = true
	return nil
}
