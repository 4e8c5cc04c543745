//go:build mi355x

// Package cluster (addition): the in-node half of ClusterNode.SearchPoints (cluster/actions.go:275-379).  Between
// servers the reference's msgpack RPC stays as it is.  Inside one 8 x MI355X server the shards of a collection are
// pinned one per GPU; instead of eight RPCSearchPoints calls and a host-side sort, each shard's goroutine makes ONE
// call that searches its shard, exchanges the per-shard top-k blocks with a single RCCL all-gather over xGMI and
// merges them on its GPU (sdb_cluster_search_batch).  Every GPU holds the merged answer; shard 0's goroutine alone
// copies it to the host.
//
// Order.  REST requests are concurrent (httpapi/v2/handlers.go:435-489) and each fans out to its shards from
// goroutines of its own (actions.go:316-351), so without further care shard 0 may see request A before B and shard 1
// B before A -- harmless over RPC, fatal for a collective, whose all-gathers would pair A's block with B's.  Every
// request therefore draws ONE ticket (1, 2, 3, ... from a counter shared by all ranks) and hands it to every rank's
// call; the library lets a rank's calls into the exchange in ticket order whichever goroutine arrives first, stamps
// each block with (sequence, ticket, shape, hash of the queries) and refuses to merge blocks whose stamps differ
// (SDB_ERR_STATE on every rank, no answer) -- see include/semadb_amd.h "Collective calls, order and failure".  Up to
// SDB_CLUSTER_RING requests are in flight per rank: the exchange of one runs under the graph walk of the next.
// C++ twin, compiled and run on the GPU by tests/host/test_host.cpp: semadb::cluster::GpuFanout (semadb_host.hpp).
package cluster

/*
#cgo CFLAGS: -I${SRCDIR}/../third_party/semadb_amd/include
#cgo LDFLAGS: -lsemadb_amd
#include "semadb_amd.h"
*/
import "C"

import (
	"fmt"
	"runtime"
	"sync"
	"unsafe"
)

// gpuFanout owns one sdb_cluster rank per local shard (one Go process drives the whole node).
type gpuFanout struct {
	ranks   []*C.sdb_cluster
	indexes []*C.sdb_index // the shard indexes, in rank order (device r)
	mu      sync.Mutex     // guards ticket
	ticket  uint64         // the last ticket handed out; every request takes the next one, for all ranks
}

func fanoutErr(what string, rc C.int) error {
	return fmt.Errorf("%s: %s (status %d)", what, C.GoString(C.sdb_last_error()), int(rc))
}

// newGpuFanout sets up the communicator for n shards on devices 0..n-1 (ncclCommInitAll inside).  A deployment
// with one process per GPU calls sdb_cluster_unique_id on one of them, ships the 128 bytes over the existing
// cluster RPC and has every process call sdb_cluster_create instead.
func newGpuFanout(indexes []*C.sdb_index) (*gpuFanout, error) {
	n := len(indexes)
	f := &gpuFanout{ranks: make([]*C.sdb_cluster, n), indexes: indexes}
	if rc := C.sdb_cluster_create_local(C.int(n), nil, &f.ranks[0]); rc != C.SDB_OK {
		return nil, fanoutErr("could not create the shard exchange", rc)
	}
	var next C.uint64_t
	if rc := C.sdb_cluster_next_ticket(f.ranks[0], &next); rc != C.SDB_OK {
		f.close()
		return nil, fanoutErr("could not read the exchange's ticket counter", rc)
	}
	f.ticket = uint64(next) - 1
	return f, nil
}

// nextTicket: one per request, without gaps.  A ticket that never reaches some rank would hold up every later one
// there until the library's deadline (sdb_cluster_set_deadline) fails them one by one, so searchPoints below presents
// its ticket to every rank unconditionally (a cancelled request still runs) and, should a rank's goroutine die before
// its call, gives the ticket up for that rank with sdb_cluster_skip_ticket -- the reference fails one request and
// serves the next (actions.go:339-353).
func (f *gpuFanout) nextTicket() uint64 {
	f.mu.Lock()
	defer f.mu.Unlock()
	f.ticket++
	return f.ticket
}

func (f *gpuFanout) close() {
	for _, r := range f.ranks {
		C.sdb_cluster_destroy(r)
	}
}

// searchPoints answers nq queries (row-major, dim floats each) over all local shards.  limit is the request's
// original limit; the per-shard limit of actions.go:291-299 is applied inside.  Returns ids[nq*limit] (shard-local
// node ids), shards[nq*limit] (which shard each id belongs to), dists and counts.
func (f *gpuFanout) searchPoints(queries []float32, nq, limit, searchSize int) (ids []uint64, shards []uint32, dists []float32, counts []uint32, err error) {
	n := len(f.ranks)
	if nq <= 0 || limit <= 0 || len(queries) == 0 { // nothing to ask: no ticket is drawn (and &queries[0] would panic)
		return nil, nil, nil, make([]uint32, 0), nil
	}
	ticket := f.nextTicket()
	// every rank's GPU ends up with the same merged answer: rank 0 alone copies it to the host, the other ranks take
	// part in the exchange with NULL outputs and return its verdict
	ids, shards = make([]uint64, nq*limit), make([]uint32, nq*limit)
	dists, counts = make([]float32, nq*limit), make([]uint32, nq)
	rcs := make([]C.int, n)
	msgs := make([]string, n)
	var wg sync.WaitGroup
	for r := 0; r < n; r++ { // the call is collective: one goroutine (OS thread while in C) per shard
		wg.Add(1)
		go func(r int) {
			defer wg.Done()
			runtime.LockOSThread() // sdb_last_error is thread-local: the call and the read of its message share a thread
			defer runtime.UnlockOSThread()
			presented := false
			defer func() {
				// whatever went wrong before the library saw the ticket on this rank: stand in for it with an empty
				// answer under an error flag, so that neither the peers (inside the all-gather) nor this rank's later
				// requests (at the turnstile) wait for it
				if rec := recover(); rec != nil || !presented {
					rcs[r] = C.sdb_cluster_skip_ticket(f.ranks[r], C.uint64_t(ticket), C.uint64_t(nq), 0, C.uint32_t(limit))
					msgs[r] = fmt.Sprintf("request abandoned on this shard: %v", rec)
					if rcs[r] == C.SDB_OK {
						rcs[r] = C.SDB_ERR_STATE
					}
				}
			}()
			var pIds *C.uint64_t
			var pDists *C.float
			var pShards, pCounts *C.uint32_t
			if r == 0 {
				pIds, pDists = (*C.uint64_t)(unsafe.Pointer(&ids[0])), (*C.float)(unsafe.Pointer(&dists[0]))
				pShards, pCounts = (*C.uint32_t)(unsafe.Pointer(&shards[0])), (*C.uint32_t)(unsafe.Pointer(&counts[0]))
			}
			presented = true
			rcs[r] = C.sdb_cluster_search_batch(f.ranks[r], f.indexes[r], C.uint64_t(ticket), C.uint64_t(nq),
				(*C.float)(unsafe.Pointer(&queries[0])), C.uint32_t(limit), C.uint32_t(searchSize), pIds, pDists, pShards, pCounts,
				C.SDB_MEM_HOST, nil)
			if rcs[r] != C.SDB_OK {
				msgs[r] = C.GoString(C.sdb_last_error()) // thread-local: read on the goroutine's OS thread, right after the call
			}
		}(r)
	}
	wg.Wait()
	// a shard that failed its search has said so inside the exchange: every rank returns an error for this request
	// and the next request is served (the reference: "could not search points", actions.go:339-353)
	for r := range rcs {
		if rcs[r] != C.SDB_OK {
			return nil, nil, nil, nil, fmt.Errorf("shard %d could not search points: %s (status %d)", r, msgs[r], int(rcs[r]))
		}
	}
	return ids, shards, dists, counts, nil
}

// perShardLimit exposes the rule of actions.go:291-299 for the servers' RPC path between nodes.
func perShardLimit(limit, nShards, maxSearchLimit int) int {
	var out C.uint32_t
	C.sdb_shard_limit(C.uint32_t(limit), C.uint32_t(nShards), C.uint32_t(maxSearchLimit), &out)
	return int(out)
}
