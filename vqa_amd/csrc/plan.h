// Host-side plan object behind the opaque vqa_plan handle.
#pragma once
#include <stdint.h>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/vqa_amd.h"

struct PlanNode { int32_t sample, token, col, child0, child1, level; };

struct PlanLevel {
    int count = 0, bcount = 0;
    int half_count = 0;   // backward tiles of the level that start with a half of their parent's transposed projection
    int off_proj = 0, off_conv1 = 0, off_conv2 = 0, off_dgrad2 = 0, off_dgrad1 = 0, off_dproj = 0;
    int off_fused_fwd = 0, off_fused_bwd = 0;   // residual body as one tile (vqa_resblock_tiles)
};

// The tile lists: a growing int32 array with the parts of std::vector's interface that the scheduler
// uses, in the plan's own memory or in memory that the caller lends (pinned memory, say, so that the
// upload needs no staging copy).  resize() leaves new words unwritten.
class PlanBlob {
public:
    size_t size() const { return n_; }
    bool empty() const { return n_ == 0; }
    bool lent() const { return lent_; }
    int32_t* data() { return p_; }
    const int32_t* data() const { return p_; }
    int32_t* begin() { return p_; }
    int32_t* end() { return p_ + n_; }
    int32_t& operator[](size_t i) { return p_[i]; }
    const int32_t& operator[](size_t i) const { return p_[i]; }
    void clear() { n_ = 0; }
    void reserve(size_t cap) {
        if (cap <= cap_) return;
        // lent memory that turns out too small is given up for memory of the plan's own
        std::vector<int32_t> grown(cap);
        for (size_t i = 0; i < n_; i++) grown[i] = p_[i];
        own_.swap(grown);
        p_ = own_.data(); cap_ = own_.size(); lent_ = false;
    }
    void resize(size_t n) {
        if (n > cap_) reserve(n > 2 * cap_ ? n : 2 * cap_);
        n_ = n;
    }
    void push_back(int32_t v) {
        if (n_ == cap_) reserve(cap_ ? 2 * cap_ : 1024);
        p_[n_++] = v;
    }
    template <class It>
    void insert(const int32_t* at, It first, It last) {   // appends only: at == end()
        (void)at;
        for (; first != last; ++first) push_back((int32_t)*first);
    }
    void lend(int32_t* p, size_t cap) { p_ = p; cap_ = cap; n_ = 0; lent_ = true; }
    void unlend() { p_ = own_.data(); cap_ = own_.size(); n_ = 0; lent_ = false; }

private:
    std::vector<int32_t> own_;
    int32_t* p_ = nullptr;
    size_t n_ = 0, cap_ = 0;
    bool lent_ = false;
};

struct vqa_plan {
    int N = 0, T = 0, C = 0, V = 0;
    int M = 0;                        // images: N, or fewer when samples share images (forward only)
    bool shared = false;              // nodes interned per (image, token, inputs): PlanNode.sample is the image
    std::unordered_map<uint64_t, int> interned;
    std::vector<PlanNode> nodes;      // execution order per sample, samples in order
    std::vector<PlanNode> pending;    // parse scratch
    std::vector<int32_t> roots;       // node id of each sample's root
    std::vector<float> used;          // [N][T]
    std::vector<int32_t> bin_index, scene_index, token_counts;
    std::vector<int32_t> rep;         // scene nodes: the first scene node of the same image (the one that runs); else itself
    std::vector<int32_t> first_scene; // per image: its representative scene node, -1 if none
    int num_exec = 0;                 // nodes that run (all but the aliased scene nodes)
    std::vector<std::vector<int>> level_nodes;
    std::vector<PlanLevel> levels;
    PlanBlob blob;
    std::vector<int32_t> work;        // build scratch, kept for its capacity
    int num_scene = 0, num_unary = 0, num_binary = 0, num_levels = 0, max_level_tiles = 0;
    int slot_F = 0, slot_S0 = 0, slot_Ysc = 0, slot_Hsc = 0, slot_Y = 0, slot_H = 0, slot_P = 0, slots_fwd = 0;
    int slot_dHsc = 0;
    int slot_dZ = 0, slot_dH = 0, slot_dP = 0, slot_dFp = 0, slot_dF = 0, slot_dS0 = 0, slots_total = 0;
    bool has_separate = true, has_fused = true;   // which per-level tile lists the blob holds
    int scene_id = 0, wgrad_chunk = 0, flags = 0;  // build arguments the deferred backward part needs
    bool finished = false;            // backward part of the blob written (or none: shared images)
    int fwd_words = 0;                // blob words of the forward part
    size_t blob_bound = 0;            // upper bound of the finished blob's size
    int sec_off[VQA_SEC_COUNT] = {0};
    int sec_cnt[VQA_SEC_COUNT] = {0};
};

void vqa_set_error(const std::string& s);
// Raises a kernel's dynamic-LDS limit on the current device: once per (kernel, device), callable from
// several threads.  False, with the error message set, if HIP refuses.
bool vqa_allow_lds(const void* kernel, int bytes, const char* who);
int vqa_cu_count();
