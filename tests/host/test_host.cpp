// test_host.cpp -- the reference's own Go tests for the hot path, replayed through the C++ host mirror
// (semadb_amd/host/semadb_host.hpp) on a real GPU.  Each block names the Go test it follows.
#include <cstdio>
#include <cstdlib>
#include <random>
#include <tuple>

#include "../../semadb_amd/host/semadb_host.hpp"

using namespace semadb;

static int g_fail = 0;
#define CHECK(cond)                                                       \
  do {                                                                    \
    if (!(cond)) {                                                        \
      std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond);         \
      g_fail++;                                                           \
    }                                                                     \
  } while (0)

static models::IndexVectorVamanaParameters vamanaParams() {  // vamana_test.go:21-27
  models::IndexVectorVamanaParameters p;
  p.VectorSize = 2, p.DistanceMetric = "euclidean", p.SearchSize = 75, p.DegreeBound = 64, p.Alpha = 1.2f;
  return p;
}

static void test_distance_table() {  // distance/distance_test.go:9-39, distance_amd64_test.go:12-27
  struct Row {
    std::vector<float> x, y;
    float dot, l2;
  } table[] = {{{0, 0, 0}, {0, 0, 0}, 0, 0},
               {{1, 1}, {1, 1}, 2, 0},
               {{1, 2, 3}, {4, 5, 6}, 32, 27},
               {{-1, -2, -3}, {-4, -5, -6}, 32, 27},
               {{-1, 2, 3}, {4, -5, 6}, 4, 83}};
  distance::FloatDistFunc dot, l2, cosd, bad;
  CHECK(!distance::GetFloatDistanceFn("dot", &dot));
  CHECK(!distance::GetFloatDistanceFn("euclidean", &l2));
  CHECK(!distance::GetFloatDistanceFn("cosine", &cosd));
  CHECK((bool)distance::GetFloatDistanceFn("manhattan", &bad));  // distance.go:81
  for (auto &r : table) {
    CHECK(dot(r.x, r.y) == -r.dot);       // dotProductDistance distance.go:19-21
    CHECK(l2(r.x, r.y) == r.l2);
    CHECK(cosd(r.x, r.y) == 1 - r.dot);   // cosineDistance distance.go:23-25
  }
}

static void test_conversion() {  // conversion/keys.go, conversion.go
  uint64_t id = 0;
  auto k = conversion::NodeKey(0x0102030405060708ull, 'v');
  CHECK(k.size() == 10 && k[0] == 'n' && k[9] == 'v' && (uint8_t)k[1] == 0x08 && (uint8_t)k[8] == 0x01);
  CHECK(conversion::NodeIdFromKey(k, 'v', &id) && id == 0x0102030405060708ull);
  CHECK(!conversion::NodeIdFromKey(k, 'e', &id));
  std::vector<uint64_t> e{1, 42, 1ull << 40};
  CHECK(conversion::BytesToEdgeList(conversion::EdgeListToBytes(e)) == e);
  std::vector<float> f{1.5f, -2.25f, 1e-30f};
  CHECK(conversion::BytesToFloat32(conversion::Float32ToBytes(f.data(), f.size())) == f);
  CHECK(conversion::BytesToUint64(conversion::Uint64ToBytes(4242)) == 4242);
}

static std::vector<vamana::IndexVectorChange> detPoints(int n) {  // shard/index/dispatch_test.go:66-89
  std::vector<vamana::IndexVectorChange> pts;
  for (int i = 0; i < n; i++) {
    float fi = (float)(i + 2);
    pts.push_back({(uint64_t)(i + 2), {fi, fi + 1}});
  }
  return pts;
}

static void test_deterministic_search() {
  diskstore::MemBucket bucket;
  auto [inv, err] = vamana::NewIndexVamana("test", vamanaParams(), &bucket);
  CHECK(!err);
  CHECK(!inv->InsertUpdateDelete(detPoints(100)));
  models::SearchVectorVamanaOptions q;
  q.Vector = {42, 43}, q.SearchSize = 75, q.Limit = 10;
  auto r = inv->Search(q);  // TestSearch_Single shard/index/search_test.go:89-144
  CHECK(!r.err && r.results.size() == 10 && r.results[0].NodeId == 42 && r.set.count(42));
  q.Limit = 5;
  q.Weight = 0.5f;  // TestSearch_OrVector :414-457
  r = inv->Search(q);
  CHECK((r.set == std::set<uint64_t>{40, 41, 42, 43, 44}));
  for (auto &sr : r.results) CHECK(sr.HybridScore + sr.HybridScore == -sr.Distance);
  q.Limit = 10, q.Weight.reset();
  vamana::IndexVamana::Filter f47{47};  // TestSearch_FilterById :196-244
  r = inv->Search(q, &f47);
  CHECK(r.results.size() == 1 && r.results[0].NodeId == 47 && r.results[0].Distance == 50.0f);
  vamana::IndexVamana::Filter f5{42, 43, 44, 45, 46};  // TestSearch_FilterSpecific :246-288
  r = inv->Search(q, &f5);
  CHECK(r.results.size() == 5 && r.set == f5 && r.results[0].NodeId == 42);
  q.SearchSize = 25, q.Limit = 30;  // search.go:23-25
  CHECK((bool)inv->Search(q).err);
  // persistence: a second index over the same bucket (what a cold reader builds, manager.go:165-181)
  std::string v;
  CHECK(bucket.Get(conversion::NodeKey(42, 'v'), &v) && v.size() == 8);
  CHECK(bucket.Get(conversion::NodeKey(42, 'e'), &v) && !v.empty() && v.size() % 8 == 0);
  CHECK(bucket.Get(vamana::MAXNODEIDKEY, &v) && conversion::BytesToUint64(v) == 101);
  auto [inv2, err2] = vamana::NewIndexVamana("test", vamanaParams(), &bucket);
  CHECK(!err2 && inv2->maxNodeId() == 101);
  q.SearchSize = 75, q.Limit = 10;
  auto a = inv->Search(q), b = inv2->Search(q);
  CHECK(a.results.size() == b.results.size());
  for (size_t i = 0; i < a.results.size() && i < b.results.size(); i++)
    CHECK(a.results[i].NodeId == b.results[i].NodeId && a.results[i].Distance == b.results[i].Distance);
  CHECK(inv->SizeInMemory() > 0);
}

static void test_invalid_ids_and_empty() {
  diskstore::MemBucket bucket;
  auto [inv, err] = vamana::NewIndexVamana("test", vamanaParams(), &bucket);
  CHECK(!err);
  // Test_EmptySearch vamana_test.go:213-228
  models::SearchVectorVamanaOptions q;
  q.Vector = {0.5f, 0.5f};
  auto r = inv->Search(q);
  CHECK(!r.err && r.set.empty() && r.results.empty());
  // Test_InvalidIdInsert vamana_test.go:77-90
  CHECK((bool)inv->InsertUpdateDelete({{0, {0.5f, 0.5f}}}));
  CHECK((bool)inv->InsertUpdateDelete({{1, {0.5f, 0.5f}}}));
  auto bad = vamanaParams();
  bad.Alpha = 3.0f;  // models/index.go:305-307
  CHECK((bool)vamana::NewIndexVamana("bad", bad, nullptr).second);
  bad = vamanaParams();
  bad.DistanceMetric = "hamming";
  CHECK((bool)vamana::NewIndexVamana("bad", bad, nullptr).second);
}

static void test_self_retrieval_and_concurrency() {
  // Test_Search vamana_test.go:230-252 (randPoints :48-61), then the same queries from 32 threads at once
  std::mt19937 rng(7);
  std::uniform_real_distribution<float> U(0, 1);
  std::vector<vamana::IndexVectorChange> pts;
  for (int i = 0; i < 200; i++) pts.push_back({(uint64_t)(i + 2), {U(rng), U(rng)}});
  diskstore::MemBucket bucket;
  auto [inv, err] = vamana::NewIndexVamana("test", vamanaParams(), &bucket);
  CHECK(!err && !inv->InsertUpdateDelete(pts));
  std::vector<std::vector<uint64_t>> seq(200);
  for (int i = 0; i < 200; i++) {
    models::SearchVectorVamanaOptions q;
    q.Vector = pts[i].Vector;
    auto r = inv->Search(q);
    CHECK(!r.err && r.results.size() == 10 && r.results[0].NodeId == pts[i].Id && r.results[0].Distance == 0);
    for (auto &sr : r.results) seq[i].push_back(sr.NodeId);
  }
  const uint64_t before = inv->deviceBatches();
  inv->setBatching(1024, std::chrono::microseconds(2000));
  std::vector<std::thread> th;
  std::atomic<int> bad{0};
  for (int t = 0; t < 32; t++)
    th.emplace_back([&, t] {
      for (int i = t; i < 200; i += 32) {
        models::SearchVectorVamanaOptions q;
        q.Vector = pts[i].Vector;
        auto r = inv->Search(q);
        std::vector<uint64_t> got;
        for (auto &sr : r.results) got.push_back(sr.NodeId);
        if (r.err || got != seq[i]) bad++;
      }
    });
  for (auto &x : th) x.join();
  CHECK(bad == 0);
  const uint64_t used = inv->deviceBatches() - before;
  std::printf("concurrent: 200 Search calls from 32 threads -> %llu device batches\n", (unsigned long long)used);
  CHECK(used < 200);  // calls were coalesced
}

static void test_create_update_delete() {
  // the CUD mix of Test_ConcurrentCUD (vamana_test.go:92-140), applied in one write like a shard transaction:
  // 50 points, then insert 50 more + update 25 + delete 10; point count, no references to deleted ids
  // (checkNoReferences shard_vector_test.go:187-214), bucket in sync, a cold reader sees the same graph
  std::mt19937 rng(11);
  std::uniform_real_distribution<float> U(0, 1);
  auto rp = [&](int n, int off) {
    std::vector<vamana::IndexVectorChange> v;
    for (int i = 0; i < n; i++) v.push_back({(uint64_t)(i + off + 2), {U(rng), U(rng)}});
    return v;
  };
  diskstore::MemBucket bucket;
  auto [inv, err] = vamana::NewIndexVamana("test", vamanaParams(), &bucket);
  CHECK(!err && !inv->InsertUpdateDelete(rp(50, 0)));
  auto more = rp(50, 50);
  auto upd = rp(25, 25);
  for (int i = 0; i < 10; i++) more.push_back({(uint64_t)(i + 2), {}});  // nil vector = delete
  more.insert(more.end(), upd.begin(), upd.end());
  more.push_back({999999, {}});  // delete of a missing id: nothing to do
  CHECK(!inv->InsertUpdateDelete(more));
  size_t nv = 0, ne = 0;
  std::set<uint64_t> ids;
  bucket.ForEach([&](const std::string &k, const std::string &v) {
    uint64_t id;
    if (conversion::NodeIdFromKey(k, 'v', &id)) nv++, ids.insert(id);
    if (conversion::NodeIdFromKey(k, 'e', &id)) ne++;
    return Error();
  });
  CHECK(nv == 91 && ne == 91);  // 100 - 10 + start node
  for (int i = 0; i < 10; i++) CHECK(!ids.count(i + 2) && !inv->Exists(i + 2));
  bucket.ForEach([&](const std::string &k, const std::string &v) {
    uint64_t id;
    if (conversion::NodeIdFromKey(k, 'e', &id))
      for (uint64_t e : conversion::BytesToEdgeList(v)) CHECK(ids.count(e));  // no edge to a deleted id
    return Error();
  });
  // the updated vectors are the ones found
  for (auto &u : upd)
    if (u.Id >= 12) {
      models::SearchVectorVamanaOptions q;
      q.Vector = u.Vector;
      auto r = inv->Search(q);
      CHECK(!r.err && !r.results.empty() && r.results[0].NodeId == u.Id && r.results[0].Distance == 0);
    }
  auto [cold, err2] = vamana::NewIndexVamana("test", vamanaParams(), &bucket);
  CHECK(!err2);
  models::SearchVectorVamanaOptions q;
  q.Vector = {0.5f, 0.5f};
  auto a = inv->Search(q), b = cold->Search(q);
  CHECK(a.results.size() == b.results.size());
  for (size_t i = 0; i < a.results.size() && i < b.results.size(); i++) CHECK(a.results[i].NodeId == b.results[i].NodeId);
}


// vectorstore.New with a product quantizer (vectorstore.go:86-91), the Fit trigger after a write
// (vamana.go:257-260, product.go:175-236) and the bucket round trip of codes and centroids
// (product.go:80-86,307-320,334-373); the store-level expectations of vectorestore_test.go:112-154 hold
// at index level as "a stored point is found by its own vector".
static void test_quantized_index() {
  models::IndexVectorVamanaParameters p;
  p.VectorSize = 8, p.DistanceMetric = "euclidean", p.SearchSize = 50, p.DegreeBound = 32, p.Alpha = 1.2f;
  models::Quantizer q;
  q.Type = models::QuantizerProduct;
  CHECK((bool)q.Validate());  // product parameters not provided (quantizer.go:20-24)
  q.Product = models::ProductQuantizerParameters{256, 3, 1000};
  p.Quantizer = q;
  diskstore::MemBucket bucket;
  {
    auto bad = vamana::IndexVamana::NewIndexVamana("pq", p, &bucket);  // 8 % 3 != 0 (product.go:44-46)
    CHECK((bool)bad.second);
  }
  p.Quantizer->Product->NumSubVectors = 4;
  CHECK(!p.Quantizer->Validate());
  diskstore::MemBucket b2;
  auto [ix, err] = vamana::IndexVamana::NewIndexVamana("pq", p, &b2);
  CHECK(!err);
  ix->setFitSeed(12345);
  std::mt19937 rng(7);
  std::uniform_real_distribution<float> U(-1, 1);
  auto mk = [&](uint64_t first, int n) {
    std::vector<vamana::IndexVectorChange> ch;
    for (int i = 0; i < n; i++) {
      vamana::IndexVectorChange c;
      c.Id = first + i;
      for (int k = 0; k < 8; k++) c.Vector.push_back(U(rng));
      ch.push_back(c);
    }
    return ch;
  };
  auto a = mk(2, 700);
  CHECK(!ix->InsertUpdateDelete(a));
  CHECK(!ix->quantized());  // 701 points < TriggerThreshold
  std::string tmp;
  CHECK(!b2.Get(vamana::productQuantizerFlatCentroidsKey, &tmp));
  auto b = mk(702, 500);
  CHECK(!ix->InsertUpdateDelete(b));
  CHECK(ix->quantized());  // 1 201 points: fitted, encoded, persisted
  CHECK(b2.Get(vamana::productQuantizerFlatCentroidsKey, &tmp) && tmp.size() == 4 * 256 * 2 * 4);
  CHECK(b2.Get(vamana::productQuantizerCentroidDistsKey, &tmp) && tmp.size() == 4 * 256 * 256 * 4);
  CHECK(b2.Get(conversion::NodeKey(2, 'q'), &tmp) && tmp.size() == 4);
  CHECK(b2.Get(conversion::NodeKey(vamana::STARTID, 'q'), &tmp));  // the start node lives in the same store
  // writes keep working on the quantized store: insert, update, delete
  auto c = mk(1202, 100);
  c.push_back({5, {}});                 // delete
  c.push_back({6, a[4].Vector});        // update id 6 with id 6's old neighbour's vector
  CHECK(!ix->InsertUpdateDelete(c));
  CHECK(!ix->Exists(5) && ix->Exists(1250));
  CHECK(!b2.Get(conversion::NodeKey(5, 'q'), &tmp));
  // a stored point is retrieved by its own vector (table distances, no re-ranking)
  int found = 0;
  std::vector<std::vector<models::SearchResult>> before;
  for (int i = 0; i < 20; i++) {
    models::SearchVectorVamanaOptions so;
    so.Vector = b[i * 7].Vector, so.SearchSize = 50, so.Limit = 10;
    auto r = ix->Search(so);
    CHECK(!r.err && r.results.size() == 10);
    for (auto &x : r.results) found += x.NodeId == b[i * 7].Id;
    before.push_back(r.results);
  }
  CHECK(found >= 15);
  // cold reader: a second index over the same bucket comes up quantized and answers identically
  auto [ix2, err2] = vamana::IndexVamana::NewIndexVamana("pq", p, &b2);
  CHECK(!err2 && ix2->quantized());
  for (int i = 0; i < 20; i++) {
    models::SearchVectorVamanaOptions so;
    so.Vector = b[i * 7].Vector, so.SearchSize = 50, so.Limit = 10;
    auto r = ix2->Search(so);
    CHECK(!r.err && r.results.size() == before[i].size());
    for (size_t k = 0; k < r.results.size() && k < before[i].size(); k++)
      CHECK(r.results[k].NodeId == before[i][k].NodeId && r.results[k].Distance == before[i][k].Distance);
  }
}

// flat.IndexFlat (flat.go): insert, replace, delete and the exact scan, with the bucket written through and a cold
// reader seeing the same points (shard_vector_test.go runs its cases over both index types)
static void test_flat_index() {
  std::mt19937 rng(5);
  std::uniform_real_distribution<float> U(0, 1);
  models::IndexVectorFlatParameters p;
  p.VectorSize = 8, p.DistanceMetric = "euclidean";
  diskstore::MemBucket bucket;
  auto [ix, err] = flat::IndexFlat::NewIndexFlat(p, &bucket);
  CHECK(!err);
  std::vector<vamana::IndexVectorChange> pts;
  for (int i = 0; i < 200; i++) {
    std::vector<float> v(8);
    for (auto &x : v) x = U(rng);
    pts.push_back({(uint64_t)(i + 1), v});
  }
  CHECK(!ix->InsertUpdateDelete(pts));
  // every point finds itself at distance 0
  for (int i = 0; i < 200; i += 17) {
    models::SearchVectorFlatOptions q;
    q.Vector = pts[i].Vector, q.Limit = 3;
    auto r = ix->Search(q);
    CHECK(!r.err && r.results.size() == 3 && r.results[0].NodeId == pts[i].Id && r.results[0].Distance == 0);
    CHECK(r.results[0].HybridScore == 0 && r.results[1].HybridScore == -r.results[1].Distance);
  }
  // move point 1 onto point 2's vector shifted, delete point 2, delete a missing id
  std::vector<float> moved = pts[1].Vector;
  moved[0] += 0.001f;
  CHECK(!ix->InsertUpdateDelete({{1, moved}, {2, {}}, {777, {}}}));
  models::SearchVectorFlatOptions q;
  q.Vector = pts[1].Vector, q.Limit = 128;  // the device scan serves limits up to 128
  auto r = ix->Search(q);
  CHECK(!r.err && r.results.size() == 128 && r.results[0].NodeId == 1 && !r.set.count(2));
  // filter
  flat::IndexFlat::Filter f = {3, 4, 2, 900};
  auto rf = ix->Search(q, &f);
  CHECK(!rf.err && rf.results.size() == 2 && rf.set.count(3) && rf.set.count(4));
  // the bucket holds exactly the live points; a cold index answers the same
  size_t nv = 0;
  bucket.ForEach([&](const std::string &k, const std::string &) {
    uint64_t id;
    if (conversion::NodeIdFromKey(k, 'v', &id)) nv++;
    return Error();
  });
  CHECK(nv == 199);
  auto [cold, err2] = flat::IndexFlat::NewIndexFlat(p, &bucket);
  CHECK(!err2);
  auto rc = cold->Search(q);
  CHECK(rc.results.size() == r.results.size());
  std::set<std::pair<float, uint64_t>> a, b;  // equal as sets of (distance, id): the bucket order is the key order
  for (auto &x : r.results) a.insert({x.Distance, x.NodeId});
  for (auto &x : rc.results) b.insert({x.Distance, x.NodeId});
  CHECK(a == b);
  // wrong metric / wrong length
  models::IndexVectorFlatParameters bad = p;
  bad.DistanceMetric = "manhattan";
  CHECK((bool)flat::IndexFlat::NewIndexFlat(bad, nullptr).second);
  q.Vector.pop_back();
  CHECK((bool)ix->Search(q).err);
}

// ClusterNode.SearchPoints inside one server (cluster/actions.go:316-376) through cluster::GpuFanout, the compiled
// twin of integration/go/cluster/fanout_mi355x.go: two shards that share the GPU, requests from racing threads (one
// ticket each), every answer equal to the reference's rule applied on the host to the shards' own answers -- per-shard
// limit (:291-299), concatenate, sort by distance (HybridScore descending), truncate (:357-376).  Then a shard that
// cannot search: the request fails on every rank, nobody hangs, the next request is served.
static void test_cluster_fanout() {
  const int kShards = 2, kPer = 400, kReq = 24;
  std::vector<std::unique_ptr<vamana::IndexVamana>> shards;
  std::mt19937 rng(7);
  std::uniform_real_distribution<float> u(0, 100);
  for (int s = 0; s < kShards; s++) {
    auto [inv, err] = vamana::NewIndexVamana("shard", vamanaParams(), nullptr);
    CHECK(!err);
    std::vector<vamana::IndexVectorChange> pts;
    for (int i = 0; i < kPer; i++) pts.push_back({(uint64_t)(i + 2), {u(rng), u(rng)}});
    CHECK(!inv->InsertUpdateDelete(pts));
    shards.push_back(std::move(inv));
  }
  std::vector<sdb_index *> hs;
  for (auto &s : shards) hs.push_back(s->handle());
  auto [fan, ferr] = cluster::GpuFanout::New(hs, std::vector<int>(kShards, 0), 2);
  CHECK(!ferr);
  if (ferr) {
    std::printf("fanout: %s\n", ferr.msg.c_str());
    return;
  }
  CHECK(cluster::PerShardLimit(10, 8) == 10 && cluster::PerShardLimit(100, 5) == 38);  // actions.go:291-299
  const int nq = 8, limit = 10, L = 75;
  std::vector<std::vector<float>> batches(kReq, std::vector<float>((size_t)nq * 2));
  for (auto &b : batches)
    for (auto &x : b) x = u(rng);
  // what the reference's merge makes of the shards' own answers
  auto expect = [&](const std::vector<float> &q, std::vector<uint64_t> *ids, std::vector<uint32_t> *sh) {
    const int per = cluster::PerShardLimit(limit, kShards);
    for (int i = 0; i < nq; i++) {
      std::vector<std::tuple<float, uint32_t, uint64_t>> all;
      for (int s = 0; s < kShards; s++) {
        models::SearchVectorVamanaOptions o;
        o.Vector = {q[(size_t)2 * i], q[(size_t)2 * i + 1]}, o.SearchSize = L, o.Limit = per;
        auto r = shards[(size_t)s]->Search(o);
        CHECK(!r.err);
        for (auto &x : r.results) all.emplace_back(x.Distance, (uint32_t)s, x.NodeId);
      }
      std::sort(all.begin(), all.end());
      for (int k = 0; k < limit && k < (int)all.size(); k++) ids->push_back(std::get<2>(all[(size_t)k])), sh->push_back(std::get<1>(all[(size_t)k]));
    }
  };
  std::vector<std::vector<uint64_t>> want_ids(kReq);
  std::vector<std::vector<uint32_t>> want_sh(kReq);
  for (int b = 0; b < kReq; b++) expect(batches[(size_t)b], &want_ids[(size_t)b], &want_sh[(size_t)b]);
  std::atomic<int> bad{0}, done{0};
  std::vector<std::thread> clients;
  for (int t = 0; t < 6; t++)
    clients.emplace_back([&, t] {
      for (int b = t; b < kReq; b += 6) {
        auto r = fan->SearchPoints(batches[(size_t)b].data(), nq, limit, L);
        if (r.err) {
          bad++;
          continue;
        }
        for (int i = 0; i < nq; i++) {
          if (r.counts[(size_t)i] != (uint32_t)limit) bad++;
          for (int k = 0; k < limit; k++)
            if (r.ids[(size_t)i * limit + k] != want_ids[(size_t)b][(size_t)i * limit + k] ||
                r.shards[(size_t)i * limit + k] != want_sh[(size_t)b][(size_t)i * limit + k])
              bad++;
        }
        done++;
      }
    });
  for (auto &c : clients) c.join();
  CHECK(bad == 0 && done == kReq);
  // a shard without a start node cannot search (search.go:57-60): the request fails, the fan-out lives on
  sdb_index_params p{};
  p.dim = 2, p.metric = SDB_METRIC_EUCLIDEAN, p.search_size = 75, p.degree_bound = 64, p.alpha = 1.2f, p.device = 0;
  sdb_index *broken = nullptr;
  CHECK(sdb_index_create(&p, &broken) == SDB_OK);
  fan->setIndex(1, broken);
  auto r = fan->SearchPoints(batches[0].data(), nq, limit, L);
  CHECK((bool)r.err);
  fan->setIndex(1, hs[1]);
  r = fan->SearchPoints(batches[0].data(), nq, limit, L);
  CHECK(!r.err && r.ids == want_ids[0] && r.shards == want_sh[0]);
  fan.reset();
  sdb_index_destroy(broken);
}

// a write that cannot go through must not wedge the index (sdb_index_abort_write): a bad point after good ones, then
// the same index takes the next write
static void test_two_precision_search_switch() {
  // setTwoPrecisionSearch (SDB_TUNE_SKETCH) through the mirror: the batch walk of a cosine table answers with the same
  // bits with and without the float16 copy, the copy stays current through later writes, and its audit finds nothing
  std::mt19937 rng(11);
  std::normal_distribution<float> N01;
  const int d = 64, n = 1500;
  models::IndexVectorVamanaParameters p;
  p.VectorSize = d, p.DistanceMetric = "cosine", p.SearchSize = 40, p.DegreeBound = 32, p.Alpha = 1.2f;
  std::vector<vamana::IndexVectorChange> pts;
  for (int i = 0; i < n; i++) {
    std::vector<float> v(d);
    float s = 0;
    for (auto &x : v) x = N01(rng), s += x * x;
    for (auto &x : v) x /= std::sqrt(s);
    pts.push_back({(uint64_t)(i + 2), v});
  }
  diskstore::MemBucket bucket;
  auto [inv, err] = vamana::NewIndexVamana("test", p, &bucket);
  CHECK(!err && !inv->InsertUpdateDelete(std::vector<vamana::IndexVectorChange>(pts.begin(), pts.begin() + 1000)));
  CHECK(sdb_index_set_tuning(inv->handle(), SDB_TUNE_WIDE_WALK, 1) == SDB_OK);  // the batch walk for this small call too
  const uint32_t nq = 64, k = 10;
  std::vector<float> q(nq * d);
  for (auto &x : q) x = N01(rng);
  auto ask = [&](std::vector<uint64_t> &ids, std::vector<float> &dd, std::vector<uint32_t> &cc) {
    ids.assign(nq * k, 0), dd.assign(nq * k, 0.f), cc.assign(nq, 0);
    return sdb_index_search_batch(inv->handle(), nq, q.data(), k, 40, nullptr, nullptr, ids.data(), dd.data(), cc.data(), nullptr,
                                  SDB_MEM_HOST, nullptr);
  };
  std::vector<uint64_t> i0, i1;
  std::vector<float> d0, d1;
  std::vector<uint32_t> c0, c1;
  CHECK(ask(i0, d0, c0) == SDB_OK);
  CHECK(!inv->setTwoPrecisionSearch(true));
  CHECK(sdb_index_set_tuning(inv->handle(), SDB_TUNE_SKETCH, 2) == SDB_OK);  // (the same switch with its audit)
  uint64_t st[3] = {0, 0, 0};
  CHECK(sdb_index_sketch_stats(inv->handle(), st) == SDB_OK && st[2] == 1);
  CHECK(ask(i1, d1, c1) == SDB_OK);
  CHECK(i0 == i1 && c0 == c1 && std::memcmp(d0.data(), d1.data(), d0.size() * 4) == 0);
  CHECK(!inv->InsertUpdateDelete(std::vector<vamana::IndexVectorChange>(pts.begin() + 1000, pts.end())));  // commit keeps the copy current
  CHECK(sdb_index_sketch_stats(inv->handle(), st) == SDB_OK && st[2] == 1);
  CHECK(ask(i1, d1, c1) == SDB_OK);
  CHECK(!inv->setTwoPrecisionSearch(false));
  CHECK(ask(i0, d0, c0) == SDB_OK);
  CHECK(i0 == i1 && c0 == c1 && std::memcmp(d0.data(), d1.data(), d0.size() * 4) == 0);
  CHECK(!inv->setTwoPrecisionSearch(true));
  CHECK(sdb_index_sketch_stats(inv->handle(), st) == SDB_OK && st[1] == 0);
}

static void test_failed_write_leaves_the_index_writable() {
  auto [inv, err] = vamana::NewIndexVamana("test", vamanaParams(), nullptr);
  CHECK(!err);
  CHECK(!inv->InsertUpdateDelete(detPoints(50)));
  std::vector<vamana::IndexVectorChange> bad = detPoints(60);
  bad.erase(bad.begin(), bad.begin() + 50);  // ids 52..61: new
  bad.push_back({70, {1.0f, 2.0f, 3.0f}});   // wrong length (models/index.go:182-184)
  CHECK((bool)inv->InsertUpdateDelete(bad));
  CHECK(!inv->Exists(52));
  bad.pop_back();
  CHECK(!inv->InsertUpdateDelete(bad));  // the index was left as it was and takes the write
  CHECK(inv->Exists(52) && inv->Exists(61));
  // the C ABI itself: an explicit transaction abandoned before its first change
  sdb_index *h = inv->handle();
  CHECK(sdb_index_begin_write(h) == SDB_OK);
  CHECK(sdb_index_begin_write(h) != SDB_OK);  // already open
  CHECK(sdb_index_abort_write(h) == SDB_OK);
  CHECK(sdb_index_begin_write(h) == SDB_OK);
  CHECK(sdb_index_commit(h, nullptr) == SDB_OK);
  CHECK(sdb_index_abort_write(h) == SDB_OK);  // nothing open: fine
  // ... and after its first change: rolled back -- the point is gone, the index answers and takes the next write
  models::SearchVectorVamanaOptions q;
  q.Vector = {5, 6}, q.Limit = 3;
  auto before = inv->Search(q);
  CHECK(!before.err);
  CHECK(sdb_index_begin_write(h) == SDB_OK);
  uint64_t id = 500;
  float v[2] = {5, 6};
  CHECK(sdb_index_insert_batch(h, 1, &id, v, SDB_MEM_HOST, 0, nullptr) == SDB_OK);
  uint8_t there = 0;
  CHECK(sdb_index_exists_batch(h, 1, &id, &there) == SDB_OK && there == 1);
  CHECK(sdb_index_abort_write(h) == SDB_OK);
  CHECK(sdb_index_exists_batch(h, 1, &id, &there) == SDB_OK && there == 0);
  uint64_t diff = 1;
  CHECK(sdb_index_version_diff(h, &diff) == SDB_OK && diff == 0);
  auto after = inv->Search(q);
  CHECK(!after.err && after.results.size() == before.results.size());
  for (size_t i = 0; i < after.results.size() && i < before.results.size(); i++)
    CHECK(after.results[i].NodeId == before.results[i].NodeId && after.results[i].Distance == before.results[i].Distance);
  CHECK(!inv->InsertUpdateDelete({{500, {5.0f, 6.0f}}}));
  CHECK(inv->Exists(500));
}

int main() {
  int ndev = 0;
  if (sdb_device_count(&ndev) != SDB_OK) {
    std::printf("no GPU: %s\n", sdb_last_error());
    return 2;
  }
  test_distance_table();
  test_conversion();
  test_deterministic_search();
  test_invalid_ids_and_empty();
  test_self_retrieval_and_concurrency();
  test_create_update_delete();
  test_quantized_index();
  test_flat_index();
  test_two_precision_search_switch();
  test_failed_write_leaves_the_index_writable();
  test_cluster_fanout();
  if (g_fail) {
    std::printf("%d HOST CHECKS FAILED\n", g_fail);
    return 1;
  }
  std::printf("ALL HOST TESTS PASSED\n");
  return 0;
}
