//go:build mi355x

// Package flat: drop-in replacement of shard/index/flat with the vector store in one MI355X's HBM and the exact
// scan as a device kernel.  Same exported surface as the reference package (flat.go:17,21,33,37,41,76).  Only the
// plain store: a quantized flat index stays on the reference's CPU path (NewIndexFlat says so).
package flat

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
	"sync"
	"unsafe"

	"github.com/RoaringBitmap/roaring/roaring64"
	"github.com/semafind/semadb/conversion"
	"github.com/semafind/semadb/diskstore"
	"github.com/semafind/semadb/models"
	"github.com/semafind/semadb/shard/index/vamana"
)

// flat.go:17-19.  The reference passes IndexFlat by value (its only field is an interface); the shared state
// lives behind one pointer here for the same reason.
type IndexFlat struct {
	s *flatState
}

type flatState struct {
	dim    int
	bucket diskstore.Bucket
	h      *C.sdb_index
	mu     sync.Mutex // writers exclude each other; searches run on the last committed rows and take no lock
}

func lastErr(what string, rc C.int) error {
	return fmt.Errorf("%s: %s (status %d)", what, C.GoString(C.sdb_last_error()), int(rc))
}

var metricCode = map[string]C.uint32_t{
	models.DistanceEuclidean: C.SDB_METRIC_EUCLIDEAN,
	models.DistanceCosine:    C.SDB_METRIC_COSINE,
	models.DistanceDot:       C.SDB_METRIC_DOT,
}

// flat.go:21-32
func NewIndexFlat(params models.IndexVectorFlatParameters, bucket diskstore.Bucket) (inf IndexFlat, err error) {
	mc, ok := metricCode[params.DistanceMetric]
	if !ok { // vectorstore.New -> distance.GetFloatDistanceFn (distance.go:60-72)
		err = fmt.Errorf("failed to create vector store: unknown float32 distance function: %s", params.DistanceMetric)
		return
	}
	if params.Quantizer != nil && params.Quantizer.Type != models.QuantizerNone {
		err = fmt.Errorf("failed to create vector store: quantizer %s of a flat index is not on the MI355X path", params.Quantizer.Type)
		return
	}
	var p C.sdb_index_params
	p.dim, p.metric = C.uint32_t(params.VectorSize), mc
	p.search_size, p.degree_bound, p.alpha = 75, 64, 1.2 // unused without a graph
	s := &flatState{dim: int(params.VectorSize), bucket: bucket}
	if rc := C.sdb_index_create(&p, &s.h); rc != C.SDB_OK {
		err = lastErr("failed to create vector store", rc)
		return
	}
	if err = s.loadFromBucket(); err != nil {
		C.sdb_index_destroy(s.h)
		return
	}
	inf.s = s
	return
}

// plainPoint.ReadFrom for every point of the bucket (plain.go:125-141): the reference reads lazily through its
// item cache, the device store is filled once.
func (s *flatState) loadFromBucket() error {
	var ids []uint64
	var vecs []float32
	err := s.bucket.ForEach(func(k, val []byte) error {
		id, ok := conversion.NodeIdFromKey(k, 'v')
		if !ok {
			return nil
		}
		if len(val) != 4*s.dim {
			return fmt.Errorf("vector of point %d has %d bytes", id, len(val))
		}
		ids = append(ids, id)
		vecs = append(vecs, conversion.BytesToFloat32(val)...)
		return nil
	})
	if err != nil || len(ids) == 0 {
		return err
	}
	if rc := C.sdb_index_set_vectors(s.h, C.uint64_t(len(ids)), (*C.uint64_t)(unsafe.Pointer(&ids[0])),
		(*C.float)(unsafe.Pointer(&vecs[0])), C.SDB_MEM_HOST); rc != C.SDB_OK {
		return lastErr("failed to load vector store", rc)
	}
	return nil
}

// Close frees the device store (the reference has nothing to free).
func (inf IndexFlat) Close() {
	if inf.s != nil && inf.s.h != nil {
		C.sdb_index_destroy(inf.s.h)
		inf.s.h = nil
	}
}

// flat.go:33-35
func (inf IndexFlat) SizeInMemory() int64 {
	var b C.int64_t
	C.sdb_index_size_in_memory(inf.s.h, &b)
	return int64(b)
}

// flat.go:37-39
func (inf IndexFlat) UpdateBucket(bucket diskstore.Bucket) {
	inf.s.bucket = bucket
}

// flat.go:41-74
func (inf IndexFlat) InsertUpdateDelete(ctx context.Context, points <-chan vamana.IndexVectorChange) <-chan error {
	errC := make(chan error, 1)
	go func() {
		defer close(errC)
		if err := inf.s.insertUpdateDelete(ctx, points); err != nil {
			errC <- fmt.Errorf("failed to insert/update/delete: %w", err)
			return
		}
		errC <- nil
	}()
	return errC
}

// one run of consecutive sets or of consecutive deletes; the two kinds keep the order the caller gave them and an
// id is set at most once per run (sdb_index_set_vectors refuses a repeat: the last Set wins by call order)
type flatRun struct {
	del  bool
	ids  []uint64
	vecs []float32
	seen map[uint64]struct{}
}

func (s *flatState) insertUpdateDelete(ctx context.Context, points <-chan vamana.IndexVectorChange) error {
	s.mu.Lock()
	defer s.mu.Unlock()
	var runs []*flatRun
	for p := range points {
		del := p.Vector == nil
		if !del && len(p.Vector) != s.dim {
			return fmt.Errorf("vector of point %d has length %d, the index has %d", p.Id, len(p.Vector), s.dim)
		}
		var cur *flatRun
		if n := len(runs); n > 0 && runs[n-1].del == del {
			if _, again := runs[n-1].seen[p.Id]; del || !again {
				cur = runs[n-1]
			}
		}
		if cur == nil {
			cur = &flatRun{del: del, seen: make(map[uint64]struct{})}
			runs = append(runs, cur)
		}
		cur.ids = append(cur.ids, p.Id)
		cur.vecs = append(cur.vecs, p.Vector...)
		cur.seen[p.Id] = struct{}{}
	}
	if err := ctx.Err(); err != nil { // utils.SinkWithContext stops on a cancelled context
		return fmt.Errorf("context done while inserting: %w", err)
	}
	if len(runs) == 0 {
		return nil
	}
	// one write transaction: concurrent searches see all of the call or none of it
	if rc := C.sdb_index_begin_write(s.h); rc != C.SDB_OK {
		return lastErr("could not start the write", rc)
	}
	committed := false
	defer func() { // every error return below leaves the transaction: the next write must not find it open
		if !committed {
			C.sdb_index_abort_write(s.h)
		}
	}()
	for _, r := range runs {
		var rc C.int
		if r.del { // vecStore.Delete (flat.go:50-52); a missing id is skipped
			rc = C.sdb_index_remove_vectors(s.h, C.uint64_t(len(r.ids)), (*C.uint64_t)(unsafe.Pointer(&r.ids[0])))
		} else { // vecStore.Set (flat.go:47-49): insert or replace
			rc = C.sdb_index_set_vectors(s.h, C.uint64_t(len(r.ids)), (*C.uint64_t)(unsafe.Pointer(&r.ids[0])),
				(*C.float)(unsafe.Pointer(&r.vecs[0])), C.SDB_MEM_HOST)
		}
		if rc != C.SDB_OK {
			return lastErr("vector store write", rc)
		}
	}
	// vecStore.Flush (plain.go:150-170): plainPoint.WriteTo / DeleteFrom
	for _, r := range runs {
		for i, id := range r.ids {
			var err error
			if r.del {
				err = s.bucket.Delete(conversion.NodeKey(id, 'v'))
			} else {
				err = s.bucket.Put(conversion.NodeKey(id, 'v'), conversion.Float32ToBytes(r.vecs[i*s.dim:(i+1)*s.dim]))
			}
			if err != nil {
				return fmt.Errorf("could not flush point %d: %w", id, err)
			}
		}
	}
	if rc := C.sdb_index_commit(s.h, nil); rc != C.SDB_OK {
		return lastErr("could not commit the write", rc)
	}
	committed = true
	// rows of replaced and deleted points are tombstones until the store is compacted
	var rows, dead C.uint64_t
	if rc := C.sdb_index_row_usage(s.h, &rows, &dead); rc == C.SDB_OK && dead*4 > rows {
		if rc := C.sdb_index_compact(s.h); rc != C.SDB_OK {
			return lastErr("could not compact the store", rc)
		}
	}
	return nil
}

// flat.go:76-132.  One query per call like the reference; the device scans the whole store per call, so a caller
// with many queries (a re-ranker, a ground-truth job) should batch them through SearchBatch.
func (inf IndexFlat) Search(ctx context.Context, options models.SearchVectorFlatOptions, filter *roaring64.Bitmap) (*roaring64.Bitmap, []models.SearchResult, error) {
	var filters []*roaring64.Bitmap
	if filter != nil {
		filters = []*roaring64.Bitmap{filter}
	}
	sets, res, err := inf.SearchBatch(ctx, []models.SearchVectorFlatOptions{options}, filters)
	if err != nil {
		return nil, nil, err
	}
	return sets[0], res[0], nil
}

// SearchBatch answers many flat searches with one pass over the store.  All options share Limit; filters is nil or
// one bitmap per query.
func (inf IndexFlat) SearchBatch(ctx context.Context, options []models.SearchVectorFlatOptions, filters []*roaring64.Bitmap) ([]*roaring64.Bitmap, [][]models.SearchResult, error) {
	s := inf.s
	nq := len(options)
	if nq == 0 {
		return nil, nil, nil
	}
	limit := options[0].Limit
	queries := make([]float32, 0, nq*s.dim)
	for _, o := range options {
		if len(o.Vector) != s.dim || o.Limit != limit {
			return nil, nil, fmt.Errorf("failed to iterate over points: query shape mismatch")
		}
		queries = append(queries, o.Vector...)
	}
	var fo, fi *C.uint64_t
	var fOff, fIds []uint64
	if filters != nil {
		fOff = make([]uint64, 1, nq+1)
		for _, f := range filters {
			fIds = append(fIds, f.ToArray()...)
			fOff = append(fOff, uint64(len(fIds)))
		}
		if len(fIds) == 0 {
			fIds = append(fIds, 0)
		}
		fo, fi = (*C.uint64_t)(unsafe.Pointer(&fOff[0])), (*C.uint64_t)(unsafe.Pointer(&fIds[0]))
	}
	ids := make([]uint64, nq*limit)
	dists := make([]float32, nq*limit)
	counts := make([]uint32, nq)
	if rc := C.sdb_index_flat_search(s.h, C.uint64_t(nq), (*C.float)(unsafe.Pointer(&queries[0])), C.uint32_t(limit), fo, fi,
		(*C.uint64_t)(unsafe.Pointer(&ids[0])), (*C.float)(unsafe.Pointer(&dists[0])),
		(*C.uint32_t)(unsafe.Pointer(&counts[0])), C.SDB_MEM_HOST, nil); rc != C.SDB_OK {
		return nil, nil, lastErr("failed to iterate over points", rc)
	}
	sets := make([]*roaring64.Bitmap, nq)
	out := make([][]models.SearchResult, nq)
	for q := range options {
		var weight float32 = 1
		if options[q].Weight != nil {
			weight = *options[q].Weight
		}
		sets[q] = roaring64.New()
		out[q] = make([]models.SearchResult, int(counts[q]))
		for i := range out[q] {
			dist := dists[q*limit+i]
			out[q][i] = models.SearchResult{NodeId: ids[q*limit+i], Distance: &dist, HybridScore: -1 * weight * dist} // flat.go:111-115
			sets[q].Add(ids[q*limit+i])
		}
	}
	return sets, out, nil
}
