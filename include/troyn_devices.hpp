// troyn_devices.hpp -- batch sharding over the GPUs of one node for callers of the troyn:: C++ surface (header-only, over include/troyn.hpp).
//
// The reference has nothing here: KernelProvider::initialize is cudaSetDevice(0) (src/kernelprovider.cuh:29-33) and every object lives on device 0.
// BASELINE.json's north_star shards batches of INDEPENDENT ciphertexts over the 8 GPUs of a node: keys and tables replicated, no collective on the data
// path, xGMI only to scatter and gather the batch.  bench.py does that with one PROCESS per GPU (torch.distributed / RCCL for the barrier); this header is the
// same partitioning for a C++ program that holds all GPUs in ONE process:
//
//     troyn::KernelProvider::initialize(0);
//     troyn::DeviceGroup group(parms, troyn::DeviceGroup::allDevices());          // one SEALContext + Evaluator per device (tables uploaded to each)
//     auto rlk = group.replicate(relin_keys);                                     // the key on every device (hipMemcpyPeerAsync from where it lies)
//     auto shards = group.scatter(a), shards_b = group.scatter(b);                // contiguous ranges of the batch, one slab per device
//     group.parallel([&](size_t i) {                                              // one host thread per device, bound to it
//         auto prod = group.evaluator(i).multiplyBatch(shards[i], shards_b[i]);
//         group.evaluator(i).relinearizeInplaceBatch(prod, rlk[i]);
//         shards[i] = std::move(prod);
//     });
//     std::vector<troyn::Ciphertext> result = group.gather(shards);               // back on the first device, in batch order
//
// How it maps on the library (include/troyhip.h, "More than one GPU in one process"): HIP's current device is per host thread; a context belongs to the device
// that was current when it was created and every library call on it binds the calling thread to that device; troyhip_malloc allocates on the current
// device and troyhip_free returns a block to the device it came from; troyhip_copy_peer is hipMemcpyPeerAsync.  Every member works on its device's DEFAULT
// stream (the C++ surface has no stream argument, like the reference's): members on different devices run concurrently, members that share a device
// (allowed: `devices` may repeat an id -- what a one-GPU box can test) take turns on that device's queue.
#pragma once
#include "troyn.hpp"
#include <exception>
#include <thread>
#include <utility>

namespace troyn {

// contiguous ranges [first, first + count) of a batch over `parts` shards: the first batch % parts shards hold one item more
inline std::vector<std::pair<size_t, size_t>> shardBatch(size_t batch, size_t parts) {
    if (!parts) throw std::invalid_argument("shardBatch: no shards");
    std::vector<std::pair<size_t, size_t>> out(parts);
    size_t first = 0;
    for (size_t i = 0; i < parts; i++) {
        const size_t n = batch / parts + (i < batch % parts ? 1 : 0);
        out[i] = {first, n};
        first += n;
    }
    return out;
}

class DeviceGroup {
public:
    // one context and one evaluator per entry of `devices` (HIP device ids; an id may appear more than once).  The calling thread is left bound to devices[0],
    // the group's HOME device: where scatter() expects the batch and gather() puts the results.
    DeviceGroup(const EncryptionParameters &parms, const std::vector<int> &devices, bool expand_mod_chain = true, SecurityLevel sec = SecurityLevel::tc128) : devices_(devices) {
        if (devices.empty()) throw std::invalid_argument("DeviceGroup: no devices");
        const int count = KernelProvider::deviceCount();
        for (int d : devices)
            if (d < 0 || d >= count) throw std::invalid_argument("DeviceGroup: no such device");
        for (int d : devices) {
            KernelProvider::setDevice(d);
            contexts_.push_back(std::make_unique<SEALContext>(parms, expand_mod_chain, sec));
            evaluators_.push_back(std::make_unique<Evaluator>(*contexts_.back()));
        }
        KernelProvider::setDevice(devices_[0]);
    }
    static std::vector<int> allDevices() {
        std::vector<int> d((size_t)KernelProvider::deviceCount());
        for (size_t i = 0; i < d.size(); i++) d[i] = (int)i;
        return d;
    }
    size_t size() const { return devices_.size(); }
    int device(size_t i) const { return devices_.at(i); }
    int home() const { return devices_[0]; }
    const SEALContext &context(size_t i) const { return *contexts_.at(i); }
    const Evaluator &evaluator(size_t i) const { return *evaluators_.at(i); }
    std::vector<std::pair<size_t, size_t>> shards(size_t batch) const { return shardBatch(batch, size()); }

    // the key-switching keys on every member's device: one copy per member (a device-to-device copy where source and member share a device, a peer copy else);
    // `keys_device`: where `keys` lie (default: the home device)
    template <class K> std::vector<K> replicate(const K &keys, int keys_device = -1) const {
        const int from = keys_device < 0 ? home() : keys_device;
        sync(from);
        std::vector<K> out;
        for (size_t i = 0; i < size(); i++) {
            KernelProvider::setDevice(devices_[i]);
            out.push_back(KSwitchKeys::replicate(keys, from, devices_[i]));
        }
        KernelProvider::setDevice(home());
        return out;
    }
    // items (on the home device, one shape) -> one slab per member holding its contiguous range, bound to that member's context
    std::vector<std::vector<Ciphertext>> scatter(const std::vector<Ciphertext> &items) const {
        std::vector<std::vector<Ciphertext>> out(size());
        if (items.empty()) return out;
        sync(home()); // whatever produced the items has finished before another device's copy engine reads them
        const auto ranges = shards(items.size());
        for (size_t i = 0; i < size(); i++) {
            const size_t first = ranges[i].first, count = ranges[i].second;
            KernelProvider::setDevice(devices_[i]);
            if (!count) continue;
            out[i] = Ciphertext::allocateBatch(count, items[first]);
            move(out[i], 0, devices_[i], items, first, home(), count);
            for (Ciphertext &c : out[i]) c.bind(*contexts_[i]);
        }
        for (size_t i = 0; i < size(); i++) sync(devices_[i]);
        KernelProvider::setDevice(home());
        return out;
    }
    // the members' slabs -> ONE slab on the home device, in batch order, bound to the home context
    std::vector<Ciphertext> gather(const std::vector<std::vector<Ciphertext>> &parts) const {
        if (parts.size() != size()) throw std::invalid_argument("gather: one slab per member");
        size_t total = 0;
        const Ciphertext *like = nullptr;
        for (const auto &p : parts) { total += p.size(); if (!like && !p.empty()) like = &p[0]; }
        if (!like) return {};
        for (size_t i = 0; i < size(); i++) sync(devices_[i]); // the members' work has finished before the home device pulls the results
        KernelProvider::setDevice(home());
        std::vector<Ciphertext> out = Ciphertext::allocateBatch(total, *like);
        size_t at = 0;
        for (size_t i = 0; i < size(); i++) {
            if (parts[i].empty()) continue;
            if (!parts[i][0].sameShape(*like)) throw std::invalid_argument("gather: shards of different shape");
            move(out, at, home(), parts[i], 0, devices_[i], parts[i].size());
            at += parts[i].size();
        }
        sync(home());
        for (Ciphertext &c : out) c.bind(*contexts_[0]);
        return out;
    }
    // fn(i) for every member on a host thread of its own, bound to the member's device; returns when every member's device is idle.  The first exception
    // of any member is rethrown here.
    template <class F> void parallel(F fn) const {
        std::vector<std::exception_ptr> errors(size());
        std::vector<std::thread> threads;
        for (size_t i = 0; i < size(); i++)
            threads.emplace_back([&, i] {
                try {
                    KernelProvider::setDevice(devices_[i]);
                    fn(i);
                    check(troyhip_stream_synchronize(nullptr));
                } catch (...) { errors[i] = std::current_exception(); }
            });
        for (std::thread &t : threads) t.join();
        KernelProvider::setDevice(home());
        for (const std::exception_ptr &e : errors)
            if (e) std::rethrow_exception(e);
    }
private:
    static void sync(int device) { KernelProvider::setDevice(device); check(troyhip_stream_synchronize(nullptr)); }
    // dst[at .. at + count) <- src[first .. first + count): one peer copy when both are dense runs of a slab, one per ciphertext otherwise (issued with the
    // destination's device current, on its default stream)
    static void move(std::vector<Ciphertext> &dst, size_t at, int dst_device, const std::vector<Ciphertext> &src, size_t first, int src_device, size_t count) {
        std::vector<const Ciphertext *> s, d;
        for (size_t b = 0; b < count; b++) { s.push_back(&src[first + b]); d.push_back(&dst[at + b]); }
        const size_t words = src[first].size() * src[first].coeffModulusSize() * src[first].polyModulusDegree();
        for (size_t b = 0; b < count; b++)
            if (!src[first + b].sameShape(src[first])) throw std::invalid_argument("scatter / gather: ciphertexts of different shape");
        for (size_t b = 0; b < count; b++) dst[at + b].copyMeta(*src[first + b].raw()); // (before isBatch looks at the destination's shape)
        if (Ciphertext::isBatch(s) && Ciphertext::isBatch(d)) {
            check(troyhip_copy_peer(dst[at].raw()->data, dst_device, src[first].raw()->data, src_device, count * words * 8, nullptr));
        } else {
            for (size_t b = 0; b < count; b++) check(troyhip_copy_peer(dst[at + b].raw()->data, dst_device, src[first + b].raw()->data, src_device, words * 8, nullptr));
        }
    }
    std::vector<int> devices_;
    std::vector<std::unique_ptr<SEALContext>> contexts_;
    std::vector<std::unique_ptr<Evaluator>> evaluators_;
};

} // namespace troyn
