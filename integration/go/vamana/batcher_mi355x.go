//go:build mi355x

package vamana

/*
#include "semadb_amd.h"
*/
import "C"

import (
	"context"
	"runtime"
	"sync"
	"sync/atomic"
	"time"
	"unsafe"

	"github.com/RoaringBitmap/roaring/roaring64"
)

// One REST request carries one query (httpapi/v2/handlers.go:435-489) and the reference answers it on its own
// goroutine; a single walk leaves the GPU idle (it is ~80 dependent hops).  searchBatcher coalesces concurrent submit
// calls into sdb_index_search_batch calls with host buffers; `workers` goroutines keep that many device batches in
// flight (the library hands every call its own workspace).  semadb_host.hpp's SearchBatcher is the same logic in
// compiled form, measured by bench.py (config.batcher_qps).
//
// What a request costs on the host: ONE atomic add reserves slot i of the batch that is filling (no lock); the caller's
// goroutine copies its vector straight into that batch's pinned slab (sdb_host_alloc: the H2D copy of the device call is
// then a single DMA); the submit that takes the last slot hands the batch to the workers; a batch that does not fill
// within `window` is sealed by a worker.  Requests with the prevailing (limit, searchSize) and no filter take this
// path; filtered ones (they carry a bitmap) and other parameters go through a queue that a worker groups.
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

type fillBatch struct {
	mem     unsafe.Pointer // pinned [maxBatch][dim] float32
	queries []float32      // the same memory as a slice
	// pinned result slabs for limits up to slabLimit (one block: ids | dists | counts): with queries AND outputs
	// page-locked a device call is one kernel launch that reads and writes them in place (semadb_amd.h sdb_host_alloc)
	resMem    unsafe.Pointer
	resIds    []uint64
	resDists  []float32
	resCounts []uint32
	reqs    []*searchReq
	st      atomic.Uint64 // tagOf(key)<<32 | slots reserved; slots >= maxBatch: full or sealed
	written atomic.Uint32 // slots whose vector has been copied in
	count   uint32        // slots that belong to the batch once it is sealed
	key     uint64        // limit<<32 | searchSize of every request in it
	tFirst  atomic.Int64  // UnixNano of its first request
}

type searchBatcher struct {
	ix       *IndexVamana
	dim      int
	maxBatch int
	window   time.Duration
	cur      atomic.Pointer[fillBatch] // the batch that is filling
	mu       sync.Mutex                // sealed, queued, free, fastKey: once per BATCH on the fast path
	cond     *sync.Cond                // workers sleep here
	freeCond *sync.Cond                // submitters wait here when every slab is in use
	sealed   []*fillBatch
	queued   []*searchReq
	free     []*fillBatch
	all      []*fillBatch
	fastKey  uint64
	streak   int
	stopped  bool
	stopping atomic.Bool // the same, readable without b.mu on the fast path
	busy     atomic.Int32 // workers inside a device call
	wg       sync.WaitGroup
}

// slabLimit: the largest limit a batch's page-locked result slab holds (the API's maximum is 75, models/search.go:287-297)
const slabLimit = 128

func newSearchBatcher(ix *IndexVamana, maxBatch int, window time.Duration, workers int) *searchBatcher {
	b := &searchBatcher{ix: ix, dim: int(ix.parameters.VectorSize), maxBatch: maxBatch, window: window}
	b.cond = sync.NewCond(&b.mu)
	b.freeCond = sync.NewCond(&b.mu)
	for i := 0; i < workers+2; i++ {
		fb := &fillBatch{reqs: make([]*searchReq, maxBatch)}
		if rc := C.sdb_host_alloc(C.size_t(maxBatch*b.dim*4), &fb.mem); rc == C.SDB_OK && fb.mem != nil {
			fb.queries = unsafe.Slice((*float32)(fb.mem), maxBatch*b.dim)
		} else {
			fb.mem, fb.queries = nil, make([]float32, maxBatch*b.dim) // pageable memory works too, only slower
		}
		resBytes := maxBatch*slabLimit*12 + maxBatch*4
		if rc := C.sdb_host_alloc(C.size_t(resBytes), &fb.resMem); rc == C.SDB_OK && fb.resMem != nil {
			fb.resIds = unsafe.Slice((*uint64)(fb.resMem), maxBatch*slabLimit)
			fb.resDists = unsafe.Slice((*float32)(unsafe.Add(fb.resMem, maxBatch*slabLimit*8)), maxBatch*slabLimit)
			fb.resCounts = unsafe.Slice((*uint32)(unsafe.Add(fb.resMem, maxBatch*slabLimit*12)), maxBatch)
		} else {
			fb.resMem = nil // flushSlab then answers into Go slices: staged by the driver, same answers
		}
		b.all = append(b.all, fb)
		b.free = append(b.free, fb)
	}
	b.cur.Store(b.takeFree())
	for i := 0; i < workers; i++ {
		b.wg.Add(1)
		go b.loop()
	}
	return b
}

func (b *searchBatcher) stop() {
	b.mu.Lock()
	b.stopped = true
	b.stopping.Store(true)
	b.mu.Unlock()
	b.cond.Broadcast()
	b.freeCond.Broadcast()
	b.wg.Wait()
	for _, fb := range b.all {
		if fb.mem != nil {
			C.sdb_host_free(fb.mem)
		}
		if fb.resMem != nil {
			C.sdb_host_free(fb.resMem)
		}
	}
}

// takeFree: b.mu held (or construction)
func (b *searchBatcher) takeFree() *fillBatch {
	if len(b.free) == 0 {
		return nil
	}
	fb := b.free[len(b.free)-1]
	b.free = b.free[:len(b.free)-1]
	fb.written.Store(0)
	fb.tFirst.Store(0)
	fb.count, fb.key = 0, b.fastKey
	tag, ok := tagOf(b.fastKey)
	if !ok {
		tag = 0xFFFFFFFF // no request's tag: parameters that do not fit one fill no batch
	}
	fb.st.Store(uint64(tag) << 32)
	return fb
}

// tagOf: the 32-bit form of (limit<<32 | searchSize) that shares the slot atomic.  The batch's parameters and its
// slot counter live in ONE atomic and a slot is reserved by compare-and-swap on both, so that a slab which was sealed,
// run, recycled for other parameters and installed again between a submitter's look at the parameters and its
// reservation fails the swap instead of handing the request a slot of a batch with another (limit, searchSize).
// Parameters beyond 16 bits (the API's maxima are 75 and 75, models/search.go:287-297) take the queue.
func tagOf(key uint64) (uint32, bool) {
	limit, L := key>>32, key&0xFFFFFFFF
	if limit > 0xFFFF || L > 0xFFFF {
		return 0, false
	}
	return uint32(limit<<16 | L), true
}

// seal closes a batch to further reservations: what its counter held before is what belongs to it (>= maxBatch: it
// was full or sealed already and whoever did that is rotating it).  At most one seal per life adds to the counter.
func (b *searchBatcher) seal(fb *fillBatch) uint32 {
	for {
		st := fb.st.Load()
		if uint32(st) >= uint32(b.maxBatch) {
			return uint32(st)
		}
		if fb.st.CompareAndSwap(st, st+uint64(b.maxBatch)) {
			return uint32(st)
		}
	}
}

// rotateLocked: fb is full or was sealed with `count` reserved slots; b.mu held
func (b *searchBatcher) rotateLocked(fb *fillBatch, count uint32) {
	fb.count = count
	if count == uint32(b.maxBatch) {
		b.streak = 0
	}
	if count > 0 {
		b.sealed = append(b.sealed, fb)
	} else {
		b.free = append(b.free, fb)
	}
	b.cur.Store(b.takeFree())
	b.cond.Signal()
	b.freeCond.Broadcast()
}

// submit enqueues one query and waits for its answer or for the context.  A cancelled request is still
// answered by the device batch it is in; nobody reads the answer.
func (b *searchBatcher) submit(ctx context.Context, vector []float32, limit, searchSize int, filter *roaring64.Bitmap) ([]uint64, []float32, error) {
	r := &searchReq{vector: vector, limit: limit, searchSize: searchSize, filter: filter, done: make(chan searchResp, 1)}
	key := uint64(limit)<<32 | uint64(searchSize)
	placed := false
	tag, fast := tagOf(key)
	fast = fast && filter == nil
	for fast && !placed {
		if b.stopping.Load() { // the workers are leaving: a slot reserved now might never be run
			return nil, nil, context.Canceled
		}
		fb := b.cur.Load()
		if fb == nil { // every slab is in use (back-pressure) or a rotation is under way
			b.mu.Lock()
			for b.cur.Load() == nil && !b.stopped {
				b.freeCond.Wait()
			}
			stopped := b.stopped
			b.mu.Unlock()
			if stopped {
				return nil, nil, context.Canceled
			}
			continue
		}
		st := fb.st.Load()
		if uint32(st>>32) != tag {
			break // other parameters than the filling batch's: the queue
		}
		i := uint32(st)
		if i >= uint32(b.maxBatch) {
			// full or sealed: its last submitter / a worker is installing the next one
			for b.cur.Load() == fb && uint32(fb.st.Load()) >= uint32(b.maxBatch) {
				runtime.Gosched()
			}
			continue
		}
		if !fb.st.CompareAndSwap(st, st+1) { // slot i of THIS life of the slab, with THESE parameters, or nothing
			continue
		}
		if i == 0 {
			fb.tFirst.Store(time.Now().UnixNano())
		}
		fb.reqs[i] = r
		copy(fb.queries[int(i)*b.dim:(int(i)+1)*b.dim], vector) // into the pinned slab
		fb.written.Add(1)
		if int(i)+1 == b.maxBatch { // the submit that takes the last slot hands the batch on
			b.mu.Lock()
			b.rotateLocked(fb, uint32(b.maxBatch))
			b.mu.Unlock()
		} else if i == 0 {
			// a worker starts this batch's window.  Through the lock: a worker that has just read "no first request
			// yet" under it is inside its Wait by the time the signal is sent (a signal between its look and its Wait
			// would be lost and the batch would sit for the worker's poll instead of its window) -- found with the C++
			// twin under the stress of tests/host/test_concurrency.cpp, fixed in both
			b.mu.Lock()
			b.mu.Unlock() //nolint:staticcheck // empty critical section on purpose
			b.cond.Signal()
		}
		placed = true
	}
	if !placed {
		b.mu.Lock()
		if b.stopped {
			b.mu.Unlock()
			return nil, nil, context.Canceled
		}
		b.queued = append(b.queued, r)
		if fast { // when the unfiltered traffic has moved to other parameters the fast path follows it
			b.streak++
			if b.fastKey == 0 || b.streak > 4*b.maxBatch {
				b.fastKey, b.streak = key, 0
				if fb := b.cur.Load(); fb != nil && fb.key != key {
					if got := b.seal(fb); got < uint32(b.maxBatch) {
						b.rotateLocked(fb, got)
					}
				}
			}
		}
		b.mu.Unlock()
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
		var fb *fillBatch
		var others []*searchReq
		b.mu.Lock()
		for {
			if len(b.sealed) > 0 {
				fb, b.sealed = b.sealed[0], b.sealed[1:]
				break
			}
			if len(b.queued) > 0 {
				others, b.queued = b.queued, nil
				break
			}
			// a partial batch whose first request has waited `window`: seal it.  Adding maxBatch to its slot
			// counter closes it; what the counter held before is what belongs to it.
			c := b.cur.Load()
			var age time.Duration
			first := int64(0)
			if c != nil {
				first = c.tFirst.Load()
			}
			if first != 0 {
				age = time.Duration(time.Now().UnixNano() - first)
			}
			// ... or at once when no device batch is running: waiting then buys nothing and the window would only be
			// added to the request's latency (a lone Search call); under load arrivals pile up behind the running batch
			if first != 0 && (age >= b.window || b.stopped || b.busy.Load() == 0) {
				if got := b.seal(c); got < uint32(b.maxBatch) {
					b.rotateLocked(c, got) // ours to seal
				} else {
					// the submit that took its last slot is rotating it and needs b.mu for that: let it have it
					b.mu.Unlock()
					runtime.Gosched()
					b.mu.Lock()
				}
				continue
			}
			if b.stopped {
				b.mu.Unlock()
				return
			}
			if first != 0 { // sync.Cond has no timed wait: a timer wakes this worker when the window is over
				t := time.AfterFunc(b.window-age, b.cond.Signal)
				b.cond.Wait()
				t.Stop()
			} else {
				b.cond.Wait()
			}
		}
		b.mu.Unlock()
		if fb != nil {
			b.busy.Add(1)
			b.flushSlab(fb)
			b.busy.Add(-1)
			b.cond.Signal() // what piled up behind this batch can go now
			b.mu.Lock()
			b.free = append(b.free, fb)
			if b.cur.Load() == nil {
				b.cur.Store(b.takeFree())
			}
			b.mu.Unlock()
			b.freeCond.Broadcast()
		} else if len(others) > 0 {
			for len(others) > 0 { // one device call per (limit, searchSize, filtered) group, oldest group first
				head := others[0]
				var batch, rest []*searchReq
				for _, r := range others {
					if len(batch) < b.maxBatch && r.limit == head.limit && r.searchSize == head.searchSize && (r.filter != nil) == (head.filter != nil) {
						batch = append(batch, r)
					} else {
						rest = append(rest, r)
					}
				}
				others = rest
				b.busy.Add(1)
				b.flush(batch)
				b.busy.Add(-1)
			}
			b.cond.Signal()
		}
	}
}

// flushSlab runs a sealed batch straight from its pinned slab
func (b *searchBatcher) flushSlab(fb *fillBatch) {
	nq := int(fb.count)
	for fb.written.Load() < fb.count { // the last copies in flight
		runtime.Gosched()
	}
	limit, L := int(fb.key>>32), int(fb.key&0xFFFFFFFF)
	var ids []uint64
	var dists []float32
	var counts []uint32
	inSlab := fb.resMem != nil && limit <= slabLimit
	if inSlab { // the batch's own page-locked result slab: the kernel writes the answers in place
		ids, dists, counts = fb.resIds[:nq*limit], fb.resDists[:nq*limit], fb.resCounts[:nq]
	} else {
		ids, dists, counts = make([]uint64, nq*limit), make([]float32, nq*limit), make([]uint32, nq)
	}
	rc := C.sdb_index_search_batch(b.ix.h, C.uint64_t(nq), (*C.float)(unsafe.Pointer(&fb.queries[0])),
		C.uint32_t(limit), C.uint32_t(L), nil, nil,
		(*C.uint64_t)(unsafe.Pointer(&ids[0])), (*C.float)(unsafe.Pointer(&dists[0])),
		(*C.uint32_t)(unsafe.Pointer(&counts[0])), nil, C.SDB_MEM_HOST, nil)
	for i := 0; i < nq; i++ {
		r := fb.reqs[i]
		fb.reqs[i] = nil
		if rc != C.SDB_OK {
			r.done <- searchResp{err: lastErr("search_batch", rc)}
			continue
		}
		n := int(counts[i])
		if inSlab { // the slab is refilled by the next batch: the caller gets its own copy
			r.done <- searchResp{ids: append([]uint64(nil), ids[i*limit:i*limit+n]...), dists: append([]float32(nil), dists[i*limit:i*limit+n]...)}
		} else {
			r.done <- searchResp{ids: ids[i*limit : i*limit+n], dists: dists[i*limit : i*limit+n]}
		}
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

// bitmapsAreSmaller: would the batch's filters take fewer bytes as bitmaps over [Minimum, Maximum] than as id lists?
// True for dense filters (an inverted-index hit list over a large share of the shard): 100 000 ids out of a million are
// 800 KB as a list and 125 KB as a bitmap, and the upload is what a large filter costs.
func bitmapsAreSmaller(reqs []*searchReq) bool {
	var words, card uint64
	for _, r := range reqs {
		if r.filter.IsEmpty() {
			continue
		}
		card += r.filter.GetCardinality()
		words += (r.filter.Maximum()-(r.filter.Minimum()&^63))/64 + 1
	}
	return card > 4096 && words < card
}

// packFilterBitmaps lays the batch's roaring bitmaps out for sdb_index_search_batch_bitmap: query q's filter is
// {first[q] + i : bit i of words[offsets[q]:offsets[q+1]]}.
func packFilterBitmaps(reqs []*searchReq) (first, offsets, words []uint64) {
	offsets = make([]uint64, 1, len(reqs)+1)
	for _, r := range reqs {
		if r.filter.IsEmpty() {
			first = append(first, 0)
			offsets = append(offsets, uint64(len(words)))
			continue
		}
		f0 := r.filter.Minimum() &^ 63
		base := len(words)
		words = append(words, make([]uint64, (r.filter.Maximum()-f0)/64+1)...)
		for it := r.filter.Iterator(); it.HasNext(); {
			v := it.Next() - f0
			words[base+int(v/64)] |= 1 << (v % 64)
		}
		first = append(first, f0)
		offsets = append(offsets, uint64(len(words)))
	}
	if len(words) == 0 {
		words = append(words, 0) // a valid pointer for cgo; no query reads it
	}
	return first, offsets, words
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
	var rc C.int
	if reqs[0].filter != nil && bitmapsAreSmaller(reqs) {
		first, wOff, words := packFilterBitmaps(reqs)
		rc = C.sdb_index_search_batch_bitmap(b.ix.h, C.uint64_t(nq), (*C.float)(unsafe.Pointer(&queries[0])),
			C.uint32_t(limit), C.uint32_t(L), (*C.uint64_t)(unsafe.Pointer(&first[0])), (*C.uint64_t)(unsafe.Pointer(&wOff[0])),
			(*C.uint64_t)(unsafe.Pointer(&words[0])),
			(*C.uint64_t)(unsafe.Pointer(&ids[0])), (*C.float)(unsafe.Pointer(&dists[0])),
			(*C.uint32_t)(unsafe.Pointer(&counts[0])), nil, C.SDB_MEM_HOST, nil)
	} else {
		fOff, fIds := packFilters(reqs)
		var fo, fi *C.uint64_t
		if fOff != nil {
			fo, fi = (*C.uint64_t)(unsafe.Pointer(&fOff[0])), (*C.uint64_t)(unsafe.Pointer(&fIds[0]))
		}
		rc = C.sdb_index_search_batch(b.ix.h, C.uint64_t(nq), (*C.float)(unsafe.Pointer(&queries[0])),
			C.uint32_t(limit), C.uint32_t(L), fo, fi,
			(*C.uint64_t)(unsafe.Pointer(&ids[0])), (*C.float)(unsafe.Pointer(&dists[0])),
			(*C.uint32_t)(unsafe.Pointer(&counts[0])), nil, C.SDB_MEM_HOST, nil)
	}
	for i, r := range reqs {
		if rc != C.SDB_OK {
			r.done <- searchResp{err: lastErr("search_batch", rc)}
			continue
		}
		n := int(counts[i])
		r.done <- searchResp{ids: ids[i*limit : i*limit+n], dists: dists[i*limit : i*limit+n]}
	}
}
