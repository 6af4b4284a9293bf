// context.h -- device context of libtroyhip: the MI355X counterpart of SEALContextCuda /
// ContextDataCuda / RNSToolCuda / NTTTablesCuda (src/context_cuda.cu:5-62, src/utils/rns_cuda.cu:206-269,
// src/utils/ntt_cuda.cuh:24-29).  Built from the encryption parameters alone; every table lives in HBM
// for the lifetime of the context.
#pragma once
#include "device_types.h"
#include "hostmath.h"
#include <map>
#include <memory>
#include <vector>

namespace troyhip {

struct BehzDev; // behz.hip

enum Scheme { SCHEME_BFV = 1, SCHEME_CKKS = 2, SCHEME_BGV = 3 };

// Grow-only device scratch (the reference's DeviceDynamicArray::ensure, src/utils/devicearray.cuh:289-297).
// One arena per context; ops on one context are stream-ordered by the caller.
class Arena {
public:
    ~Arena();
    // carve `count` u64 from the arena for the current op; valid until reset()
    u64 *take(size_t count);
    void reset() { used_ = 0; }
    // every operation that carves scratch opens with begin(stream): the arena (and with it the context) belongs to ONE stream at
    // a time -- two streams on one context would silently overwrite each other's scratch, so the second one is refused until the
    // owner is released (troyhip_context_release_stream, after the caller has synchronised the first stream)
    void begin(hipStream_t s) {
        if (owned_ && owner_ != s)
            throw Error(ST_LOGIC_ERROR, "this context is in use on another stream: one context (scratch arena) per stream, or troyhip_context_release_stream after synchronising");
        owner_ = s;
        owned_ = true;
        used_ = 0;
    }
    void release_owner() { owned_ = false; }
    void reserve(size_t count);
    size_t capacity() const { return cap_; }
private:
    u64 *base_ = nullptr;
    size_t cap_ = 0, used_ = 0;
    hipStream_t owner_ = nullptr;
    bool owned_ = false;
    std::vector<u64 *> retired_; // blocks replaced by a larger one; freed with the context
};

struct Level {
    int limbs = 0;
    host::RnsLevel rns;
    std::vector<uint8_t> bsk_ids;
    std::shared_ptr<BehzDev> behz; // BFV only
    const Shoup *d_inv_qlast = nullptr; // [limbs - 1]  q_last^-1 mod q_l on the device (Ntt1Corr::inv of the CKKS rescale at this level)
    std::vector<void *> dev_blocks;
};

class Context {
public:
    // with_device = false builds the host tables only (no HIP call): enough for hostcrypto (config A plumbing)
    Context(int scheme, u64 N, const std::vector<u64> &primes, u64 t, bool with_device = true);
    bool has_device = true;
    int device = 0; // the HIP device the tables, the scratch arena and every launch of this context live on: the creating thread's current device
    ~Context();
    Context(const Context &) = delete;
    Context &operator=(const Context &) = delete;

    int scheme;
    u64 N;
    int logn;
    int K;          // key-level limb count
    u64 t;
    int first_limbs, last_limbs;
    std::vector<u64> primes;     // registry: [0,K) key primes, then the BEHZ aux primes
    std::vector<host::NttTable> tables;
    std::vector<PrimeDesc> h_desc;
    PrimeDesc *d_desc = nullptr;
    // d_desc with q_special^-1 mod q_j folded into the N^-1 constants of the data primes (and in `aux`): the inverse transform of the
    // key-switch accumulators then delivers acc qk^-1, half of the mod-down (Ntt1ModDown / Ntt2ModDown, kernels.h).  BFV and BGV contexts with a special prime.
    PrimeDesc *d_desc_md = nullptr;
    const Shoup *d_inv_qk = nullptr; // [K - 1]  q_special^-1 mod q_j on the device (Ntt1Corr::inv of the CKKS key-switch mod-down)
    std::map<int, Level> levels; // by limb count, K .. last_limbs
    Arena arena;

    // does a launch over `rows` limb rows put fewer than `per_cu` workgroup tiles (2048 coefficients) on a compute unit?  (one ciphertext, a few small ones:
    // kernels that cannot hide their own latencies -- the evaluator then merges launches that are separate at the large batch)
    bool small_launch(u64 rows, unsigned per_cu = 64) const;

    const Level &level(int limbs) const;
    bool has_level(int limbs) const { return levels.count(limbs) != 0; }
    bool is_data_level(int limbs) const { return has_level(limbs) && (K == 1 || limbs < K); }
    int prime_id(u64 p) const;
    LimbMap ct_map(int limbs) const;            // rows cycle through key primes 0..limbs-1
    LimbMap ids_map(const std::vector<uint8_t> &ids, uint32_t inner = 1) const;
    LimbMap single_map(int id) const;

private:
    int register_prime(u64 p);
    void upload_tables();
    void build_level(int limbs);
    std::vector<void *> dev_allocs_;
    template <class T> T *upload(const std::vector<T> &v, std::vector<void *> &owner);
};

} // namespace troyhip
