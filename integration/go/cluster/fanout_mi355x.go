//go:build mi355x

// Package cluster (addition): the in-node half of ClusterNode.SearchPoints (cluster/actions.go:275-379).  Between
// servers the reference's msgpack RPC stays as it is.  Inside one 8 x MI355X server the shards of a collection are
// pinned one per GPU; instead of eight RPCSearchPoints calls and a host-side sort, each shard's goroutine makes ONE
// call that searches its shard, exchanges the per-shard top-k blocks with a single RCCL all-gather over xGMI and
// merges them on its GPU (sdb_cluster_search_batch).  Every goroutine gets the merged answer; the caller takes
// shard 0's.
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

// nextTicket: one per request, without gaps -- a ticket that never reaches some rank would hold up every later one
// there, so searchPoints below presents its ticket to every rank unconditionally (a cancelled request still runs).
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
	ticket := f.nextTicket()
	type out struct {
		ids    []uint64
		shards []uint32
		dists  []float32
		counts []uint32
		rc     C.int
	}
	outs := make([]out, n)
	var wg sync.WaitGroup
	for r := 0; r < n; r++ { // the call is collective: one goroutine (OS thread while in C) per shard
		wg.Add(1)
		go func(r int) {
			defer wg.Done()
			o := &outs[r]
			o.ids, o.shards = make([]uint64, nq*limit), make([]uint32, nq*limit)
			o.dists, o.counts = make([]float32, nq*limit), make([]uint32, nq)
			o.rc = C.sdb_cluster_search_batch(f.ranks[r], f.indexes[r], C.uint64_t(ticket), C.uint64_t(nq),
				(*C.float)(unsafe.Pointer(&queries[0])), C.uint32_t(limit), C.uint32_t(searchSize), (*C.uint64_t)(unsafe.Pointer(&o.ids[0])),
				(*C.float)(unsafe.Pointer(&o.dists[0])), (*C.uint32_t)(unsafe.Pointer(&o.shards[0])),
				(*C.uint32_t)(unsafe.Pointer(&o.counts[0])), C.SDB_MEM_HOST, nil)
		}(r)
	}
	wg.Wait()
	// a shard that failed its search has said so inside the exchange: every rank returns an error for this request
	// and the next request is served (the reference: "could not search points", actions.go:339-353)
	for r := range outs {
		if outs[r].rc != C.SDB_OK {
			return nil, nil, nil, nil, fmt.Errorf("shard %d could not search points (status %d)", r, int(outs[r].rc))
		}
	}
	o := outs[0] // every rank holds the same merged answer
	return o.ids, o.shards, o.dists, o.counts, nil
}

// perShardLimit exposes the rule of actions.go:291-299 for the servers' RPC path between nodes.
func perShardLimit(limit, nShards, maxSearchLimit int) int {
	var out C.uint32_t
	C.sdb_shard_limit(C.uint32_t(limit), C.uint32_t(nShards), C.uint32_t(maxSearchLimit), &out)
	return int(out)
}
