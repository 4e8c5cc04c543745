// turnstile.h -- the host-side ORDER of the shard exchange (cluster.hip), free of HIP: which call of a rank enters the
// exchange next (tickets), which ring slot it gets, when the ranks of a shared-device group have all arrived for a
// sequence number and how one of them takes its arrival back.  Everything here is plain C++ under one mutex, so it
// compiles -- and runs under ThreadSanitizer / AddressSanitizer -- on a box without a GPU:
// tests/host/test_concurrency.cpp drives exactly this code with 8 ranks x 8 threads, and tests/host/mock_sdb.cpp
// builds its CPU stand-in of sdb_cluster_search_batch on it.
//
// The reference has no such thing: ClusterNode.SearchPoints fans a request out over RPC from one goroutine per shard
// (cluster/actions.go:316-351) and every reply names its request; a collective needs the calls in one order on every
// rank (include/semadb_amd.h "Collective calls, order and failure").
#pragma once
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <map>
#include <mutex>
#include <set>
#include <vector>

#include "../../include/semadb_amd.h"

namespace sdb {

int fail(int code, const char *fmt, ...) noexcept;  // common.h's in the library; a harness brings its own

// what the turnstile keeps per rank.  RCCL ranks lock their own mutex; the ranks of a shared-device group share the
// group's (the last rank to arrive enqueues on everybody's streams).
struct OrderState {
  int rank = 0;
  uint64_t seq = 0;          // collectives this rank has entered
  uint64_t next_ticket = 1;  // the ticket the turnstile lets in next
  bool desync = false;       // a call left between taking its sequence number and entering the exchange
  std::set<uint64_t> skipped;    // tickets the fan-out has declared lost on this rank (sdb_cluster_skip_ticket)
  uint32_t deadline_ms = 30000;  // longest wait at the turnstile / for the peers (0: for ever)
  unsigned next_slot = 0;
  std::mutex own_mu;
  std::condition_variable own_cv;
  std::mutex *mu = &own_mu;
  std::condition_variable *cv = &own_cv;
};

// the flags of one ring slot the order depends on (all under the rank's lock)
struct SlotState {
  bool used = false;     // its `done` event has been recorded at least once
  bool busy = false;     // a host-memory call still reads its staging
  bool pending = false;  // shared transport: registered, the last rank has not enqueued it yet
};

// the ticket turnstile: a call enters in ticket order and ALWAYS gives the turn on when it leaves the locked section,
// whatever happened in between -- otherwise its successors would wait forever.  A ticket that is never presented on
// this rank (the fan-out's thread died, the request was cancelled before this rank was called) would wedge them all
// the same: the wait has a deadline (sdb_cluster_set_deadline) after which the call fails having done NOTHING -- it
// may be presented again --, and sdb_cluster_skip_ticket lets the fan-out declare a ticket lost.  The reference fails
// one request and serves the next (cluster/actions.go:339-353).
struct Turn {
  OrderState *c;
  uint64_t ticket;
  bool mine = false;
  static void advance(OrderState *c, uint64_t to) {  // lock held
    c->next_ticket = to;
    for (auto it = c->skipped.find(c->next_ticket); it != c->skipped.end(); it = c->skipped.find(c->next_ticket)) {
      c->skipped.erase(it);
      c->next_ticket++;
    }
    c->cv->notify_all();
  }
  int enter(std::unique_lock<std::mutex> &lk) {
    if (!ticket) return SDB_OK;
    if (ticket < c->next_ticket)
      return fail(SDB_ERR_INVALID, "ticket %llu has already entered the exchange on rank %d (next is %llu)",
                  (unsigned long long)ticket, c->rank, (unsigned long long)c->next_ticket);
    if (c->skipped.count(ticket))
      return fail(SDB_ERR_INVALID, "ticket %llu was skipped on rank %d (sdb_cluster_skip_ticket)", (unsigned long long)ticket, c->rank);
    // (a skip of this very ticket while it waits ends the wait too: the turn has passed over it)
    auto ready = [&] { return c->next_ticket >= ticket; };
    if (c->deadline_ms == 0) {
      c->cv->wait(lk, ready);
    } else if (!c->cv->wait_for(lk, std::chrono::milliseconds(c->deadline_ms), ready)) {
      return fail(SDB_ERR_STATE, "ticket %llu waited %u ms on rank %d for ticket %llu, which has not been presented to this rank; "
                  "the call did nothing (present it again once the missing ticket has been presented or skipped: sdb_cluster_skip_ticket)",
                  (unsigned long long)ticket, c->deadline_ms, c->rank, (unsigned long long)c->next_ticket);
    }
    if (c->next_ticket != ticket)
      return fail(SDB_ERR_INVALID, "ticket %llu was skipped on rank %d while it waited (sdb_cluster_skip_ticket)",
                  (unsigned long long)ticket, c->rank);
    mine = true;
    return SDB_OK;
  }
  // give the turn on before the end of the scope (a host-memory call, once its block is in the exchange)
  void pass() {  // lock held
    if (mine) {
      mine = false;
      advance(c, ticket + 1);
    }
  }
  ~Turn() { pass(); }  // runs with the lock held (declared after the lock)
};

// sdb_cluster_skip_ticket(nq == 0): no rank has entered the request's exchange or will; lock held
inline int skip_unentered(OrderState *c, uint64_t ticket) {
  if (ticket < c->next_ticket)
    return fail(SDB_ERR_INVALID, "ticket %llu has already entered the exchange on rank %d (next is %llu)",
                (unsigned long long)ticket, c->rank, (unsigned long long)c->next_ticket);
  if (ticket == c->next_ticket) Turn::advance(c, ticket + 1);
  else c->skipped.insert(ticket);
  return SDB_OK;
}

// a free ring slot (may wait for one); lock held.  In ring order: the slot taken is the one whose last exchange lies
// furthest back, so that waiting for it to finish only ever blocks a caller that has N exchanges in flight.
template <class Slot, int N>
Slot *take_slot(OrderState *c, Slot (&ring)[N], std::unique_lock<std::mutex> &lk) {
  for (;;) {
    for (int k = 0; k < N; k++) {
      Slot &s = ring[(c->next_slot + k) % N];
      if (s.busy || s.pending) continue;
      c->next_slot = (c->next_slot + k + 1) % N;
      return &s;
    }
    c->cv->wait(lk);
  }
}

// the rendezvous of a shared-device group: sequence number -> arrivals so far.  `A` carries at least `.owner` (the
// rank's OrderState) and `.slot` (its SlotState).
template <class A>
struct GroupState {
  std::mutex mu;
  std::condition_variable cv;
  int world = 0, device = 0, alive = 0;
  int gone = -1;  // a rank of the group that has been destroyed: no exchange can complete any more
  std::map<uint64_t, std::vector<A>> rv;

  // room for a sequence number's arrivals BEFORE the number is taken: nothing between taking the number and
  // registering the arrival may need host memory (a rank that took a number and then stayed out is out of step)
  void reserve(uint64_t seq) { rv[seq].reserve((size_t)world); }
  // register; true when this was the last rank: `all` then holds every rank's arrival and the entry is gone
  bool arrive(uint64_t seq, const A &a, std::vector<A> *all) {
    auto &arr = rv[seq];
    arr.push_back(a);
    if ((int)arr.size() != world) return false;
    all->swap(arr);
    rv.erase(seq);
    return true;
  }
  // the peers never presented this request: take rank `c`'s arrival back.  Nothing of it is on any stream yet, so if
  // it was the rank's latest sequence number the handle is exactly where it was before the call; otherwise the rank
  // is out of step.
  void withdraw(uint64_t seq, OrderState *c) {
    auto it = rv.find(seq);
    if (it != rv.end()) {
      auto &arr = it->second;
      arr.erase(std::remove_if(arr.begin(), arr.end(), [&](const A &x) { return x.owner == c; }), arr.end());
      if (arr.empty()) rv.erase(it);
    }
    if (c->seq == seq + 1) c->seq = seq;
    else c->desync = true;
  }
};

}  // namespace sdb
