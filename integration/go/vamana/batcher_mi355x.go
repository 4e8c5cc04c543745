//go:build mi355x

package vamana

/*
#include "semadb_amd.h"
*/
import "C"

import (
	"context"
	"sync"
	"time"
	"unsafe"

	"github.com/RoaringBitmap/roaring/roaring64"
)

// One REST request carries one query (httpapi/v2/handlers.go:435-489) and the reference answers it on its own
// goroutine; a single walk leaves the GPU idle (it is ~80 dependent hops).  searchBatcher collects concurrent
// submit calls that share (limit, searchSize, filtered?) for at most `window` or until `maxBatch` are waiting and
// runs them as one sdb_index_search_batch with host buffers; `workers` goroutines keep that many device batches in
// flight (the library hands every call its own workspace).  semadb_host.hpp's SearchBatcher is the same logic in
// compiled form.
type searchResp struct {
	ids   []uint64
	dists []float32
	err   error
}

type searchReq struct {
	vector     []float32
	limit      int
	searchSize int
	filter     *roaring64.Bitmap
	done       chan searchResp // buffered: a caller that gave up (context) never blocks the batcher
}

type searchBatcher struct {
	ix       *IndexVamana
	dim      int
	maxBatch int
	window   time.Duration
	mu       sync.Mutex
	cond     *sync.Cond
	queue    []*searchReq
	stopped  bool
	wg       sync.WaitGroup
}

func newSearchBatcher(ix *IndexVamana, maxBatch int, window time.Duration, workers int) *searchBatcher {
	b := &searchBatcher{ix: ix, dim: int(ix.parameters.VectorSize), maxBatch: maxBatch, window: window}
	b.cond = sync.NewCond(&b.mu)
	for i := 0; i < workers; i++ {
		b.wg.Add(1)
		go b.loop()
	}
	return b
}

func (b *searchBatcher) stop() {
	b.mu.Lock()
	b.stopped = true
	b.mu.Unlock()
	b.cond.Broadcast()
	b.wg.Wait()
}

// submit enqueues one query and waits for its answer or for the context.  A cancelled request is still
// answered by the device batch it is in; nobody reads the answer.
func (b *searchBatcher) submit(ctx context.Context, vector []float32, limit, searchSize int, filter *roaring64.Bitmap) ([]uint64, []float32, error) {
	r := &searchReq{vector: vector, limit: limit, searchSize: searchSize, filter: filter, done: make(chan searchResp, 1)}
	b.mu.Lock()
	if b.stopped {
		b.mu.Unlock()
		return nil, nil, context.Canceled
	}
	b.queue = append(b.queue, r)
	wake := len(b.queue) == 1 || len(b.queue) >= b.maxBatch
	b.mu.Unlock()
	if wake {
		b.cond.Signal()
	}
	select {
	case resp := <-r.done:
		return resp.ids, resp.dists, resp.err
	case <-ctx.Done():
		return nil, nil, ctx.Err()
	}
}

func (b *searchBatcher) loop() {
	defer b.wg.Done()
	for {
		b.mu.Lock()
		for len(b.queue) == 0 && !b.stopped {
			b.cond.Wait()
		}
		if b.stopped && len(b.queue) == 0 {
			b.mu.Unlock()
			return
		}
		if len(b.queue) < b.maxBatch { // a short window for more callers
			b.mu.Unlock()
			time.Sleep(b.window)
			b.mu.Lock()
		}
		if len(b.queue) == 0 { // another worker took them
			b.mu.Unlock()
			continue
		}
		// one device call per (limit, searchSize, filtered) group, oldest group first
		head := b.queue[0]
		var batch, rest []*searchReq
		for _, r := range b.queue {
			if len(batch) < b.maxBatch && r.limit == head.limit && r.searchSize == head.searchSize && (r.filter != nil) == (head.filter != nil) {
				batch = append(batch, r)
			} else {
				rest = append(rest, r)
			}
		}
		b.queue = rest
		more := len(rest) > 0
		b.mu.Unlock()
		if more {
			b.cond.Signal()
		}
		b.flush(batch)
	}
}

// packFilters lays the roaring bitmaps of a batch out as the C ABI wants them: per query the ids in
// ascending order (roaring iterates ascending), offsets[nq+1].  (nil, nil) when the batch is unfiltered.
func packFilters(reqs []*searchReq) (offsets, ids []uint64) {
	if reqs[0].filter == nil {
		return nil, nil
	}
	offsets = make([]uint64, 1, len(reqs)+1)
	for _, r := range reqs {
		ids = append(ids, r.filter.ToArray()...)
		offsets = append(offsets, uint64(len(ids)))
	}
	if len(ids) == 0 {
		ids = append(ids, 0) // a valid pointer for cgo; no query reads it
	}
	return offsets, ids
}

func (b *searchBatcher) flush(reqs []*searchReq) {
	nq, d := len(reqs), b.dim
	queries := make([]float32, nq*d)
	for i, r := range reqs {
		copy(queries[i*d:], r.vector)
	}
	limit, L := reqs[0].limit, reqs[0].searchSize
	ids := make([]uint64, nq*limit)
	dists := make([]float32, nq*limit)
	counts := make([]uint32, nq)
	fOff, fIds := packFilters(reqs)
	var fo, fi *C.uint64_t
	if fOff != nil {
		fo, fi = (*C.uint64_t)(unsafe.Pointer(&fOff[0])), (*C.uint64_t)(unsafe.Pointer(&fIds[0]))
	}
	rc := C.sdb_index_search_batch(b.ix.h, C.uint64_t(nq), (*C.float)(unsafe.Pointer(&queries[0])),
		C.uint32_t(limit), C.uint32_t(L), fo, fi,
		(*C.uint64_t)(unsafe.Pointer(&ids[0])), (*C.float)(unsafe.Pointer(&dists[0])),
		(*C.uint32_t)(unsafe.Pointer(&counts[0])), nil, C.SDB_MEM_HOST, nil)
	for i, r := range reqs {
		if rc != C.SDB_OK {
			r.done <- searchResp{err: lastErr("search_batch", rc)}
			continue
		}
		n := int(counts[i])
		r.done <- searchResp{ids: ids[i*limit : i*limit+n], dists: dists[i*limit : i*limit+n]}
	}
}
