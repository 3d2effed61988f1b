// Merge of per-shard top-k lists into the global top-k (the "host-side merge" of the 8-way sharded retrieve, BASELINE.json north_star;
// the reference intended faiss' IndexShards for it: retriever/retrieval_utils.py:164-182, dead code there).
//
// Every shard r hands rank 0 its own top-k of every query: scores fp32 [nq, k] descending, ids int64 [nq, k] (global ids, -1 = missing,
// missing entries last).  Shards are contiguous row ranges in rank order and each list is ordered (score desc, row position asc), so
// "shard asc, then list position asc" among equal scores IS "global row position asc" - the tie rule of the single-index search
// (FlatIPIndex.search, oracle/retrieval_ref.py: flat_ip_search).  Result: D fp32 [nq, k] descending, I int64 [nq, k], -1 / -inf padded.
//
// Two forms behind the C ABI:
//   cldrd_merge_topk        HOST pointers, host threads: a k-way merge per query (world heads, k pops); a list that turns out not to be
//                           sorted is handled by a partial sort of that query's candidates instead (same order, any input)
//   cldrd_merge_topk_device DEVICE pointers: builds (score, position) candidate lists for cldrd_topk_sort (one bitonic sort per query,
//                           world * k <= 8192) and maps the winning positions back to ids - used when the shard lists were gathered
//                           over RCCL and never left HBM
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

int cldrd_set_error(const char* msg);
extern "C" int cldrd_topk_sort(const int* counts, const int* cand_rows, const float* cand_scores, int nq, int cap, int k, float* D, int* I,
                               void* stream);

namespace {

// total order on floats, larger score first: the key the device sort uses (topk.hip: orderable); NaN sorts by its bit pattern
inline uint32_t desc_key(float s) {
    s += 0.0f;                                            // -0.0 -> +0.0: equal scores, one key
    uint32_t u;
    memcpy(&u, &s, 4);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);      // ascending orderable
    return ~u;                                            // descending
}

struct Cand {
    uint32_t key;       // desc_key(score)
    uint32_t pos;       // shard * k + position in the shard's list
};

}  // namespace

extern "C" int cldrd_merge_topk(const float* const* shard_scores, const long long* const* shard_ids, int world, long long nq, int k_in,
                                int k_out, float* D, long long* I, int nthreads) {
    if (world <= 0 || nq < 0 || k_in <= 0 || k_out <= 0 || !shard_scores || !shard_ids || (nq > 0 && (!D || !I))) {
        cldrd_set_error("merge_topk: bad arguments");
        return 1;
    }
    if ((long long)world * k_in > 0x7fffffffLL) { cldrd_set_error("merge_topk: world * k too large"); return 1; }
    for (int r = 0; r < world; ++r)
        if (nq > 0 && (!shard_scores[r] || !shard_ids[r])) { cldrd_set_error("merge_topk: null shard list"); return 1; }
    int nt = nthreads > 0 ? nthreads : (int)std::thread::hardware_concurrency();
    if (nt < 1) nt = 1;
    if (nt > 64) nt = 64;
    if ((long long)nt > nq) nt = nq > 0 ? (int)nq : 1;
    auto work = [&](int t) {
        const long long lo = nq * t / nt, hi = nq * (t + 1) / nt;
        std::vector<int> head((size_t)world), len((size_t)world);
        std::vector<Cand> all;
        for (long long q = lo; q < hi; ++q) {
            float* Dq = D + q * k_out;
            long long* Iq = I + q * k_out;
            // valid prefix of every list (entries before the first missing id) and whether it is sorted
            bool sorted = true;
            for (int r = 0; r < world; ++r) {
                const float* s = shard_scores[r] + q * k_in;
                const long long* id = shard_ids[r] + q * k_in;
                int n = 0;
                while (n < k_in && id[n] >= 0) ++n;
                for (int j = n; j < k_in; ++j) sorted &= id[j] < 0;                   // a valid entry behind a missing one: not a list we merge
                for (int j = 1; j < n; ++j) sorted &= desc_key(s[j - 1]) <= desc_key(s[j]);
                len[(size_t)r] = n;
                head[(size_t)r] = 0;
            }
            int out = 0;
            if (sorted) {
                // k-way merge: smallest (key, shard) among the heads; world is small (8), a linear scan beats a heap
                while (out < k_out) {
                    int best = -1;
                    uint32_t bk = 0;
                    for (int r = 0; r < world; ++r) {
                        if (head[(size_t)r] >= len[(size_t)r]) continue;
                        const uint32_t key = desc_key(shard_scores[r][q * k_in + head[(size_t)r]]);
                        if (best < 0 || key < bk) { best = r; bk = key; }
                    }
                    if (best < 0) break;
                    const int j = head[(size_t)best]++;
                    Dq[out] = shard_scores[best][q * k_in + j];
                    Iq[out] = shard_ids[best][q * k_in + j];
                    ++out;
                }
            } else {
                all.clear();
                for (int r = 0; r < world; ++r)
                    for (int j = 0; j < k_in; ++j)
                        if (shard_ids[r][q * k_in + j] >= 0) all.push_back({desc_key(shard_scores[r][q * k_in + j]), (uint32_t)(r * k_in + j)});
                const size_t take = std::min(all.size(), (size_t)k_out);
                std::partial_sort(all.begin(), all.begin() + (long)take, all.end(),
                                  [](const Cand& a, const Cand& b) { return a.key != b.key ? a.key < b.key : a.pos < b.pos; });
                for (size_t i = 0; i < take; ++i) {
                    const int r = (int)(all[i].pos / (uint32_t)k_in), j = (int)(all[i].pos % (uint32_t)k_in);
                    Dq[out] = shard_scores[r][q * k_in + j];
                    Iq[out] = shard_ids[r][q * k_in + j];
                    ++out;
                }
            }
            for (; out < k_out; ++out) { Dq[out] = -INFINITY; Iq[out] = -1; }
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nt; ++t) th.emplace_back(work, t);
    work(0);
    for (auto& x : th) x.join();
    return 0;
}

// ---- device form ---------------------------------------------------------------------------------------------------------------
// scores [world, nq, k_in] fp32, ids [world, nq, k_in] int64 (the layout dist.gather leaves: one [nq, k_in] block per rank).
// cand_scores / cand_pos: scratch [nq, world * k_in]; counts: scratch int32 [nq]; I32: scratch int32 [nq, k_out].
namespace {

__global__ void merge_prep_kernel(const float* __restrict__ scores, const long long* __restrict__ ids, int world, long long nq, int k_in,
                                  float* __restrict__ cand_scores, int* __restrict__ cand_pos, int* __restrict__ counts) {
    const long long q = blockIdx.x;
    const int cap = world * k_in;
    for (int c = threadIdx.x; c < cap; c += blockDim.x) {
        const int r = c / k_in, j = c - r * k_in;
        const size_t src = ((size_t)r * nq + q) * k_in + j;
        const bool ok = ids[src] >= 0;
        // a missing entry gets the key that sorts behind everything (score -inf, position 0xffffffff) and comes out as missing
        cand_scores[(size_t)q * cap + c] = ok ? scores[src] + 0.0f : -__builtin_inff();      // (+ 0.0f: -0.0 and 0.0 tie)
        cand_pos[(size_t)q * cap + c] = ok ? c : -1;
    }
    if (threadIdx.x == 0) counts[q] = cap;
}

__global__ void merge_ids_kernel(const long long* __restrict__ ids, const int* __restrict__ I32, int world, long long nq, int k_in, int k_out,
                                 long long* __restrict__ I, float* __restrict__ D) {
    const long long q = blockIdx.x;
    for (int i = threadIdx.x; i < k_out; i += blockDim.x) {
        const int c = I32[(size_t)q * k_out + i];
        long long id = -1;
        if (c >= 0) {
            const int r = c / k_in, j = c - r * k_in;
            id = ids[((size_t)r * nq + q) * k_in + j];
        } else {
            D[(size_t)q * k_out + i] = -__builtin_inff();
        }
        I[(size_t)q * k_out + i] = id;
    }
}

}  // namespace

extern "C" size_t cldrd_merge_topk_device_workspace(int world, long long nq, int k_in, int k_out) {
    if (world <= 0 || nq <= 0 || k_in <= 0 || k_out <= 0) return 0;
    const size_t cap = (size_t)world * k_in;
    return (size_t)nq * cap * 8 + (size_t)nq * 4 + (size_t)nq * k_out * 4 + 256;
}

extern "C" int cldrd_merge_topk_device(const float* scores, const long long* ids, int world, long long nq, int k_in, int k_out, float* D,
                                       long long* I, void* workspace, size_t workspace_bytes, void* stream) {
    if (world <= 0 || nq <= 0 || k_in <= 0 || k_out <= 0 || !scores || !ids || !D || !I || !workspace) {
        cldrd_set_error("merge_topk_device: bad arguments");
        return 1;
    }
    const long long cap = (long long)world * k_in;
    if (cap > 8192) { cldrd_set_error("merge_topk_device: world * k must be <= 8192 (use cldrd_merge_topk on the host)"); return 1; }
    if (nq > 0x7fffffffLL) { cldrd_set_error("merge_topk_device: too many queries"); return 1; }
    if (workspace_bytes < cldrd_merge_topk_device_workspace(world, nq, k_in, k_out)) { cldrd_set_error("merge_topk_device: workspace too small"); return 1; }
    char* w = (char*)workspace;
    float* cand_scores = (float*)w;                       w += (size_t)nq * cap * 4;
    int* cand_pos = (int*)w;                              w += (size_t)nq * cap * 4;
    int* counts = (int*)w;                                w += ((size_t)nq * 4 + 255) / 256 * 256;
    int* I32 = (int*)w;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(merge_prep_kernel, dim3((unsigned)nq), dim3(256), 0, st, scores, ids, world, nq, k_in, cand_scores, cand_pos, counts);
    // one bitonic sort per query over (score desc, position asc); entries past the candidates come out as (-inf, -1)
    const int k_sort = (int)std::min<long long>(k_out, cap);
    if (k_sort == k_out) {
        if (int rc = cldrd_topk_sort(counts, cand_pos, cand_scores, (int)nq, (int)cap, k_out, D, I32, stream)) return rc;
    } else {
        // k_out > world * k_in: sort what there is into a [nq, cap] prefix layout is not what D's stride is; pad through a second pass
        cldrd_set_error("merge_topk_device: k_out > world * k_in");
        return 1;
    }
    hipLaunchKernelGGL(merge_ids_kernel, dim3((unsigned)nq), dim3(256), 0, st, ids, I32, world, nq, k_in, k_out, I, D);
    if (hipGetLastError() != hipSuccess) { cldrd_set_error("merge_topk_device: launch failed"); return 1; }
    return 0;
}
