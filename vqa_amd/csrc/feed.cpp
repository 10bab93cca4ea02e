// Host side of the feature feed: gathers the feature rows of a batch out of a table in host memory
// (a memory-mapped shard) into one contiguous buffer (pinned, about to cross PCIe).  The reference
// loads one file per image and stacks them in the loader's collate step (DataLoader.py:118-119);
// here a batch is 1024 rows of 401 408 bytes at scattered offsets of one mapping, and a single
// thread copying them (6 GB/s) is seven times slower than the PCIe transfer that follows.
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/vqa_amd.h"
#include "plan.h"

extern "C" int vqa_gather_rows(const void* table, int64_t table_rows, int64_t row_bytes, const int64_t* index,
                               int64_t n, void* out, int32_t threads) {
    if (!table || !out || (!index && n > 0) || table_rows < 0 || row_bytes <= 0 || n < 0) {
        vqa_set_error("vqa_gather_rows: bad argument");
        return VQA_EINVAL;
    }
    for (int64_t i = 0; i < n; i++)
        if (index[i] < 0 || index[i] >= table_rows) {
            vqa_set_error("vqa_gather_rows: row " + std::to_string(index[i]) + " outside the table of " +
                          std::to_string(table_rows) + " rows");
            return VQA_EINVAL;
        }
    if (n == 0) return VQA_OK;
    // work is cut into pieces of at most 1 MB so that a few rows still spread over all threads
    const int64_t piece = row_bytes < (1 << 20) ? row_bytes : (1 << 20);
    const int64_t per_row = (row_bytes + piece - 1) / piece;
    const int64_t pieces = n * per_row;
    int64_t nt = threads < 1 ? 1 : threads;
    if (nt > pieces) nt = pieces;
    auto work = [=](int64_t t) {
        const int64_t lo = pieces * t / nt, hi = pieces * (t + 1) / nt;
        for (int64_t q = lo; q < hi; q++) {
            const int64_t r = q / per_row, off = (q % per_row) * piece;
            const int64_t len = off + piece <= row_bytes ? piece : row_bytes - off;
            std::memcpy((char*)out + r * row_bytes + off, (const char*)table + index[r] * row_bytes + off, (size_t)len);
        }
    };
    if (nt == 1) {
        work(0);
        return VQA_OK;
    }
    std::vector<std::thread> pool;
    pool.reserve((size_t)nt - 1);
    try {
        for (int64_t t = 1; t < nt; t++) pool.emplace_back(work, t);
    } catch (...) {
        // thread creation failed part-way: finish what was started, do the rest here
        const int64_t started = (int64_t)pool.size();
        for (auto& th : pool) th.join();
        for (int64_t t = started + 1; t < nt; t++) work(t);
        work(0);
        return VQA_OK;
    }
    work(0);
    for (auto& th : pool) th.join();
    return VQA_OK;
}
