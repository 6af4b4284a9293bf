// troyn_linear.hpp -- the reference's BFV / BGV linear-layer helpers (app/LinearHelper.cuh: Plain2d, Cipher2d, MatmulHelper,
// Conv2dHelper) over the troyn:: mirror in troyn.hpp, i.e. over libtroyhip.so.  Same namespace, class and member names, argument
// order, block search, coefficient layout and exceptions, so code written against the reference's app header compiles against this
// one and streams written by one side are read by the other.  (The CKKS twin of this header is troyn_app.hpp.)
//
// The packing: integers mod t sit in the COEFFICIENTS of a plaintext polynomial (BatchEncoder::encodePolynomial).  A weight block
// of inputBlock x outputBlock entries is laid out so that the negacyclic product x(X) w(X) carries <x, w[:, j]> at coefficient
// j * inputBlock + inputBlock - 1 of its batch row's stride; the GPU work is multiplyPlain (or multiply) + add.  With packLwe the
// inputBlock-th coefficients of up to inputBlock result ciphertexts are folded into one ciphertext by the field trace
// (negacyclicShift, divideByPolyModulusDegree, fieldTraceInplace: src/evaluator_cuda.cu:2251-2276, 2342-2351), which is the
// key-switch loop of the hot path.
#pragma once
#include "troyn.hpp"
#include <cassert>
#include <cmath>
#include <functional>
#include <iostream>

namespace LinearHelper {

template <typename T> inline void savet(std::ostream &stream, const T *obj) { stream.write(reinterpret_cast<const char *>(obj), sizeof(T)); }
template <typename T> inline void loadt(std::istream &stream, T *obj) { stream.read(reinterpret_cast<char *>(obj), sizeof(T)); }

inline static size_t ceilDiv(size_t a, size_t b) { return (a + b - 1) / b; }

class Cipher2d;

class Plain2d { // app/LinearHelper.cuh:21-40
public:
    std::vector<std::vector<troyn::Plaintext>> data;
    std::vector<troyn::Plaintext> &operator[](size_t id) { return data[id]; }
    const std::vector<troyn::Plaintext> &operator[](size_t id) const { return data[id]; }
    Plain2d() {}
    inline Cipher2d encrypt(const troyn::Encryptor &encryptor) const;
};

class Cipher2d { // app/LinearHelper.cuh:42-209
    using Plaintext = troyn::Plaintext;
    using Ciphertext = troyn::Ciphertext;

    template <class F> void each(F f) {
        for (auto &row : data)
            for (auto &ct : row) f(ct);
    }
    // this (op) x over two grids of one shape; the shape check and message of the reference
    template <class X, class F> void zip(const X &x, F f) {
        if (data.size() != x.data.size()) throw std::invalid_argument("Size incorrect.");
        for (size_t i = 0; i < data.size(); i++) {
            if (data[i].size() != x[i].size()) throw std::invalid_argument("Size incorrect.");
            for (size_t j = 0; j < data[i].size(); j++) f(data[i][j], x[i][j]);
        }
    }
    void read(std::istream &stream, const std::function<void(Ciphertext &)> &one) {
        size_t rows = 0, cols = 0;
        loadt(stream, &rows);
        loadt(stream, &cols);
        data.assign(rows, std::vector<Ciphertext>());
        for (auto &row : data) {
            row.resize(cols);
            for (auto &ct : row) one(ct);
        }
    }

public:
    std::vector<std::vector<Ciphertext>> data;
    std::vector<Ciphertext> &operator[](size_t id) { return data[id]; }
    const std::vector<Ciphertext> &operator[](size_t id) const { return data[id]; }
    Cipher2d() {}

    // rows, columns (size_t each), then the ciphertexts row-major in Ciphertext::save format; an empty grid writes nothing
    void save(std::ostream &stream) const {
        const size_t rows = data.size();
        if (!rows) return;
        const size_t cols = data.front().size();
        for (const auto &row : data)
            if (row.size() != cols) throw std::invalid_argument("Not a rectangle Conv2d.");
        savet(stream, &rows);
        savet(stream, &cols);
        for (const auto &row : data)
            for (const auto &ct : row) ct.save(stream);
    }
    void load(std::istream &stream) {
        read(stream, [&](Ciphertext &ct) { ct.load(stream); });
    }
    void load(std::istream &stream, const troyn::SEALContext &context) {
        read(stream, [&](Ciphertext &ct) { ct.load(stream, context); });
    }

    // every ciphertext of the grid, row-major
    std::vector<Ciphertext *> all() {
        std::vector<Ciphertext *> v;
        each([&](Ciphertext &ct) { v.push_back(&ct); });
        return v;
    }
    // The reference loops over the grid one ciphertext at a time (LinearHelper.cuh:104-140); here the whole grid is ONE batched library call per
    // operation (troyn::Evaluator::...InplaceBatch: the ciphertexts are gathered into a device slab once and stay its members) -- same limbs.
    void modSwitchToNext(const troyn::Evaluator &evaluator) { evaluator.modSwitchToNextInplaceBatch(all()); }
    void relinearize(const troyn::Evaluator &evaluator, const troyn::RelinKeys &rlk) { evaluator.relinearizeInplaceBatch(all(), rlk); }
    void switch_key(const troyn::Evaluator &evaluator, const troyn::KSwitchKeys &ksk) { evaluator.applyKeySwitchingInplaceBatch(all(), ksk); }
    void multiplyScalarInplace(const troyn::BatchEncoder &encoder, const troyn::Evaluator &evaluator, uint64_t scalar) {
        Plaintext constant;
        encoder.encodePolynomial(std::vector<uint64_t>{scalar}, constant);
        evaluator.multiplyPlainInplaceBatch(all(), constant);
    }
    void addInplace(const troyn::Evaluator &evaluator, const Cipher2d &x) {
        std::vector<Ciphertext *> mine;
        std::vector<const Ciphertext *> theirs;
        zip(x, [&](Ciphertext &c, const Ciphertext &o) { mine.push_back(&c); theirs.push_back(&o); }); // the reference's shape checks
        evaluator.addInplaceBatch(mine, theirs);
    }
    void addPlainInplace(const troyn::Evaluator &evaluator, const Plain2d &x) {
        zip(x, [&](Ciphertext &c, const Plaintext &p) { evaluator.addPlainInplace(c, p); });
    }
    Cipher2d addPlain(const troyn::Evaluator &evaluator, const Plain2d &x) const {
        Cipher2d sum = *this;
        sum.addPlainInplace(evaluator, x);
        return sum;
    }
};

inline Cipher2d Plain2d::encrypt(const troyn::Encryptor &encryptor) const {
    Cipher2d ret;
    ret.data.resize(data.size());
    for (size_t i = 0; i < data.size(); i++)
        for (const auto &p : data[i]) ret[i].push_back(encryptor.encryptSymmetric(p));
    return ret;
}

namespace detail {
// ret[b][o] = sum_i product(b, i, o), accumulated in the reference's order (i = 0 moved, i = 1, 2, .. added in place: LinearHelper.cuh:392-470), with
// ONE library call per input block i: the products of step i for every (b, o) are independent ciphertext operations of one
// shape, so they go through the batched forms (troyn::Evaluator::multiplyBatch / multiplyPlainBatch) and are added into the accumulators as a
// batch -- the limbs are those of the ciphertext-by-ciphertext loop.
//   left(b, i) / right(i, o): the two factors of product (b, i, o)
template <class L, class R> inline Cipher2d accumulateCipher(const troyn::Evaluator &evaluator, size_t rows, size_t outputs, size_t inputs, L left, R right) {
    Cipher2d ret;
    ret.data.resize(rows);
    for (auto &r : ret.data) r.resize(outputs);
    if (!rows || !outputs) return ret;
    std::vector<troyn::Ciphertext> acc;
    std::vector<const troyn::Ciphertext *> x(rows * outputs), y(rows * outputs);
    for (size_t i = 0; i < inputs; i++) {
        for (size_t b = 0; b < rows; b++)
            for (size_t o = 0; o < outputs; o++) { x[b * outputs + o] = &left(b, i); y[b * outputs + o] = &right(i, o); }
        std::vector<troyn::Ciphertext> prod = evaluator.multiplyBatch(x, y);
        if (i == 0) acc = std::move(prod);
        else evaluator.addInplaceBatch(acc, prod);
    }
    for (size_t b = 0; b < rows; b++)
        for (size_t o = 0; o < outputs; o++) ret[b][o] = std::move(acc[b * outputs + o]);
    return ret;
}
// ciphertext x plaintext, product (b, i, o) = ct(b, i, o) * pt(b, i, o): the ciphertexts that meet ONE plaintext go through one batched multiplyPlain --
// over the batch rows b when the plaintext is a weight (overRows: pt does not depend on b), over the outputs o when the plaintexts are the inputs
template <class C, class P> inline Cipher2d accumulatePlain(const troyn::Evaluator &evaluator, size_t rows, size_t outputs, size_t inputs, bool overRows, C ct, P pt) {
    Cipher2d ret;
    ret.data.resize(rows);
    for (auto &r : ret.data) r.resize(outputs);
    const size_t count = overRows ? rows : outputs, others = overRows ? outputs : rows;
    std::vector<const troyn::Ciphertext *> column(count);
    for (size_t i = 0; i < inputs && count; i++)
        for (size_t j = 0; j < others; j++) {
            for (size_t k = 0; k < count; k++) column[k] = overRows ? &ct(k, i, j) : &ct(j, i, k);
            std::vector<troyn::Ciphertext> prod = evaluator.multiplyPlainBatch(column, overRows ? pt(0, i, j) : pt(j, i, 0));
            for (size_t k = 0; k < count; k++) {
                troyn::Ciphertext &d = overRows ? ret[k][j] : ret[j][k];
                if (i == 0) d = std::move(prod[k]);
                else evaluator.addInplace(d, prod[k]);
            }
        }
    return ret;
}
} // namespace detail

// y = x W mod t for x [batchSize][inputDims], W [inputDims][outputDims] (row-major), app/LinearHelper.cuh:228-751.
// objective: 0 = the inputs travel encrypted, 1 = the weights do, 2 = both plus the weight gradient's inputs (what the block search
// minimises); packLwe: results are folded inputBlock-to-one before they travel back
class MatmulHelper {
    using Plaintext = troyn::Plaintext;
    using Ciphertext = troyn::Ciphertext;
    using GaloisKeys = troyn::GaloisKeys;

    size_t batchSize, inputDims, outputDims;
    size_t slotCount;
    size_t batchBlock, inputBlock, outputBlock;
    int objective;
    bool packLwe;

    // the number of ciphertexts that travel for blocks (b, i, o), LinearHelper.cuh:242-306
    size_t traffic(size_t b, size_t i, size_t o) const {
        const size_t bc = ceilDiv(batchSize, b), ic = ceilDiv(inputDims, i), oc = ceilDiv(outputDims, o);
        if (objective < 0 || objective > 2) throw std::runtime_error("MatmulHelper: invalid objective");
        if (!packLwe) return objective == 0 ? bc * (ic + oc) : objective == 1 ? (bc + ic) * oc : bc * inputDims + (bc + ic) * oc;
        const size_t packed = ceilDiv(bc * oc, i);
        return (objective == 0 ? bc * ic : objective == 1 ? oc * ic : bc * ic + oc * ic) + packed;
    }
    void determineBlock() {
        size_t best = 2147483647;
        batchBlock = inputBlock = outputBlock = 0;
        auto consider = [&](size_t b, size_t i, size_t o) {
            const size_t c = traffic(b, i, o);
            if (c < best) { best = c; batchBlock = b; inputBlock = i; outputBlock = o; }
        };
        if (!packLwe) { // large batch blocks first; an input block has to leave room for at least one output
            for (size_t b = batchSize; b >= 1; b--) {
                if (b >= slotCount || ceilDiv(batchSize, b) * 2 > best) continue;
                for (size_t i = 1; i < slotCount / b && i <= inputDims; i++) {
                    const size_t o = std::min(slotCount / b / i, outputDims);
                    if (o) consider(b, i, o);
                }
            }
            return;
        }
        // packing folds inputBlock results into one: a power of two near N^(1/3) (the field trace needs a power of two)
        const double target = std::pow((double)slotCount, 0.33);
        size_t i = 1;
        while (i * 2 < target) i *= 2;
        if (i > inputDims)
            for (i = 1; i < inputDims;) i *= 2;
        for (size_t b = 1; b <= batchSize && b <= slotCount; b++) {
            const size_t o = std::min(slotCount / b / i, outputDims);
            if (o) consider(b, i, o);
        }
    }

    size_t batchBlocks() const { return ceilDiv(batchSize, batchBlock); }
    size_t inputBlocks() const { return ceilDiv(inputDims, inputBlock); }
    size_t outputBlocks() const { return ceilDiv(outputDims, outputBlock); }
    size_t rowStride() const { return inputBlock * outputBlock; } // coefficients per batch row inside a polynomial
    // coefficient of a result polynomial holding (row r of its batch block, output k of its output block); `lane` = inputBlock - 1
    // straight out of the product, the slot inside the packed group after packOutputs
    size_t resultCoeff(size_t r, size_t k, size_t lane) const { return r * rowStride() + k * inputBlock + lane; }
    // visit the (batch block, output block) tiles in the order the result ciphertexts are numbered
    template <class F> void tiles(F f) const { // f(di, dj, li, ui, lj, uj)
        for (size_t di = 0, li = 0; li < batchSize; di++, li += batchBlock)
            for (size_t dj = 0, lj = 0; lj < outputDims; dj++, lj += outputBlock)
                f(di, dj, li, std::min(li + batchBlock, batchSize), lj, std::min(lj + outputBlock, outputDims));
    }
    std::vector<size_t> requiredTerms(size_t li, size_t ui, size_t lj, size_t uj) const {
        std::vector<size_t> terms;
        terms.reserve((ui - li) * (uj - lj));
        for (size_t i = li; i < ui; i++)
            for (size_t j = lj; j < uj; j++) terms.push_back(resultCoeff(i - li, j - lj, inputBlock - 1));
        return terms;
    }
    size_t packedCount() const { return ceilDiv(batchBlocks() * outputBlocks(), inputBlock); }
    void checkOperands(size_t aRows, size_t wRows) const {
        if (aRows != batchBlocks()) throw std::invalid_argument("Input batchsize incorrect.");
        if (wRows != inputBlocks()) throw std::invalid_argument("Weight input dimension incorrect.");
    }

public:
    MatmulHelper(size_t batchSize, size_t inputDims, size_t outputDims, size_t slotCount, int objective = 0, bool packLwe = true)
        : batchSize(batchSize), inputDims(inputDims), outputDims(outputDims), slotCount(slotCount), objective(objective), packLwe(packLwe) {
        determineBlock();
    }

    // weights[i][j] of block (bi, bj) -> coefficient (j - lj) * inputBlock + (inputBlock - 1) - (i - li): input i of the block, which
    // sits at X^(i - li) of the input polynomial, meets it at exponent (j - lj) * inputBlock + inputBlock - 1
    Plain2d encodeWeights(troyn::BatchEncoder &encoder, const uint64_t *weights) {
        Plain2d encoded;
        encoded.data.assign(inputBlocks(), std::vector<Plaintext>(outputBlocks()));
        std::vector<uint64_t> poly(rowStride());
        for (size_t bi = 0; bi < inputBlocks(); bi++)
            for (size_t bj = 0; bj < outputBlocks(); bj++) {
                std::fill(poly.begin(), poly.end(), 0);
                const size_t li = bi * inputBlock, lj = bj * outputBlock;
                for (size_t i = li; i < std::min(li + inputBlock, inputDims); i++)
                    for (size_t j = lj; j < std::min(lj + outputBlock, outputDims); j++) {
                        const size_t r = (j - lj) * inputBlock + (inputBlock - 1) - (i - li);
                        assert(r < slotCount);
                        poly[r] = weights[i * outputDims + j];
                    }
                encoder.encodePolynomial(poly, encoded[bi][bj]);
            }
        return encoded;
    }

    // batch block x input block -> one polynomial: row r of the block starts at coefficient r * inputBlock * outputBlock
    Plain2d encodeInputs(troyn::BatchEncoder &encoder, const uint64_t *inputs) {
        Plain2d ret;
        ret.data.assign(batchBlocks(), std::vector<Plaintext>(inputBlocks()));
        std::vector<uint64_t> poly(slotCount);
        for (size_t bb = 0; bb < batchBlocks(); bb++)
            for (size_t bi = 0; bi < inputBlocks(); bi++) {
                std::fill(poly.begin(), poly.end(), 0);
                const size_t li = bb * batchBlock, lj = bi * inputBlock;
                for (size_t i = li; i < std::min(li + batchBlock, batchSize); i++)
                    for (size_t j = lj; j < std::min(lj + inputBlock, inputDims); j++) poly[(i - li) * rowStride() + (j - lj)] = inputs[i * inputDims + j];
                encoder.encodePolynomial(poly, ret[bb][bi]);
            }
        return ret;
    }

    Cipher2d encryptInputs(const troyn::Encryptor &encryptor, troyn::BatchEncoder &encoder, const uint64_t *inputs) {
        return encodeInputs(encoder, inputs).encrypt(encryptor);
    }

    // ret[b][j] = sum_i a[b][i] * w[i][j]: encrypted inputs x plain weights, both encrypted, plain inputs x encrypted weights
    Cipher2d matmul(const troyn::Evaluator &evaluator, const Cipher2d &a, const Plain2d &w) {
        checkOperands(a.data.size(), w.data.size());
        return detail::accumulatePlain(evaluator, a.data.size(), outputBlocks(), w.data.size(), true,
                                       [&](size_t b, size_t i, size_t) -> const Ciphertext & { return a[b][i]; }, [&](size_t, size_t i, size_t o) -> const Plaintext & { return w[i][o]; });
    }
    Cipher2d matmulCipher(const troyn::Evaluator &evaluator, const Cipher2d &a, const Cipher2d &w) {
        checkOperands(a.data.size(), w.data.size());
        return detail::accumulateCipher(evaluator, a.data.size(), outputBlocks(), w.data.size(),
                                        [&](size_t b, size_t i) -> const Ciphertext & { return a[b][i]; }, [&](size_t i, size_t o) -> const Ciphertext & { return w[i][o]; });
    }
    Cipher2d matmulReverse(const troyn::Evaluator &evaluator, const Plain2d &a, const Cipher2d &w) {
        checkOperands(a.data.size(), w.data.size());
        return detail::accumulatePlain(evaluator, a.data.size(), outputBlocks(), w.data.size(), false,
                                       [&](size_t, size_t i, size_t o) -> const Ciphertext & { return w[i][o]; }, [&](size_t b, size_t i, size_t) -> const Plaintext & { return a[b][i]; });
    }

    // a bias / expected output in the layout of matmul's result (of packOutputs' result with packLwe): only the coefficients
    // decryptOutputs reads are set
    Plain2d encodeOutputs(troyn::BatchEncoder &encoder, const uint64_t *outputs) {
        Plain2d ret;
        if (!packLwe) {
            ret.data.assign(batchBlocks(), std::vector<Plaintext>(outputBlocks()));
            std::vector<uint64_t> poly(slotCount);
            tiles([&](size_t di, size_t dj, size_t li, size_t ui, size_t lj, size_t uj) {
                std::fill(poly.begin(), poly.end(), 0);
                for (size_t i = li; i < ui; i++)
                    for (size_t j = lj; j < uj; j++) poly[resultCoeff(i - li, j - lj, inputBlock - 1)] = outputs[i * outputDims + j];
                encoder.encodePolynomial(poly, ret[di][dj]);
            });
            return ret;
        }
        std::vector<std::vector<uint64_t>> packed(packedCount(), std::vector<uint64_t>(slotCount, 0));
        tiles([&](size_t di, size_t dj, size_t li, size_t ui, size_t lj, size_t uj) {
            const size_t id = di * outputBlocks() + dj;
            for (size_t i = li; i < ui; i++)
                for (size_t j = lj; j < uj; j++) packed[id / inputBlock][resultCoeff(i - li, j - lj, id % inputBlock)] = outputs[i * outputDims + j];
        });
        ret.data.assign(1, std::vector<Plaintext>(packed.size()));
        for (size_t k = 0; k < packed.size(); k++) encoder.encodePolynomial(packed[k], ret[0][k]);
        return ret;
    }

    std::vector<uint64_t> decryptOutputs(troyn::BatchEncoder &encoder, troyn::Decryptor &decryptor, const Cipher2d &outputs) {
        std::vector<uint64_t> dec(batchSize * outputDims);
        Plaintext pt;
        if (!packLwe) {
            std::vector<uint64_t> coeffs;
            tiles([&](size_t di, size_t dj, size_t li, size_t ui, size_t lj, size_t uj) {
                decryptor.decrypt(outputs[di][dj], pt);
                encoder.decodePolynomial(pt, coeffs);
                coeffs.resize(slotCount, 0);
                for (size_t i = li; i < ui; i++)
                    for (size_t j = lj; j < uj; j++) dec[i * outputDims + j] = coeffs[resultCoeff(i - li, j - lj, inputBlock - 1)];
            });
            return dec;
        }
        std::vector<std::vector<uint64_t>> packed(outputs[0].size());
        for (size_t k = 0; k < packed.size(); k++) {
            decryptor.decrypt(outputs[0][k], pt);
            encoder.decodePolynomial(pt, packed[k]);
            packed[k].resize(slotCount, 0);
        }
        tiles([&](size_t di, size_t dj, size_t li, size_t ui, size_t lj, size_t uj) {
            const size_t id = di * outputBlocks() + dj;
            for (size_t i = li; i < ui; i++)
                for (size_t j = lj; j < uj; j++) dec[i * outputDims + j] = packed[id / inputBlock][resultCoeff(i - li, j - lj, id % inputBlock)];
        });
        return dec;
    }

    // fold the results inputBlock-to-one (LinearHelper.cuh:564-623).  Per ciphertext: X^-(inputBlock - 1) brings the result coefficients
    // to multiples of inputBlock, the field trace over the log2(inputBlock) automorphisms X -> X^(N / 2^k + 1) removes every other
    // coefficient (the factor it multiplies by is divided out in front), X^slot moves the survivors to the group's free lane.
    Cipher2d packOutputs(const troyn::Evaluator &evaluator, const GaloisKeys &autoKey, const Cipher2d &cipher) {
        if (!packLwe) throw std::invalid_argument("PackLWE not enabled");
        Cipher2d ret;
        ret.data.emplace_back();
        if (cipher.data.empty() || cipher.data[0].empty()) return ret;
        const size_t lanes = inputBlock;
        size_t keep_log = 0;
        while ((size_t(1) << keep_log) != slotCount / lanes) keep_log++;
        // The reference folds one result ciphertext at a time (shift, divide, log2(inputBlock) key switches, shift, add).  The ciphertexts are
        // independent until the final additions, so here every step runs over ALL of them as one batched call: log2(inputBlock) key-switch
        // launches in total instead of per ciphertext.  Each packed ciphertext still receives its lanes in the order 0, 1, ..: same limbs.
        std::vector<const Ciphertext *> sources;
        for (const auto &row : cipher.data)
            for (size_t j = 0; j < cipher.data[0].size(); j++) sources.push_back(&row[j]);
        std::vector<Ciphertext> one = Ciphertext::packBatch(sources);
        const std::vector<Ciphertext *> all = Ciphertext::pointers(one);
        if (lanes > 1) evaluator.negacyclicShiftInplaceBatch(all, 2 * slotCount - (lanes - 1));
        evaluator.divideByPolyModulusDegreeInplaceBatch(all, slotCount / lanes);
        evaluator.fieldTraceInplaceBatch(all, autoKey, keep_log);
        const size_t total = one.size(), groups = ceilDiv(total, lanes);
        std::vector<Ciphertext *> acc;
        for (size_t lane = 0; lane < lanes && lane < total; lane++) {
            std::vector<Ciphertext *> members;
            for (size_t g = 0; g * lanes + lane < total; g++) members.push_back(&one[g * lanes + lane]);
            if (lane == 0) { acc = members; continue; }
            evaluator.negacyclicShiftInplaceBatch(members, lane);
            evaluator.addInplaceBatch(std::vector<Ciphertext *>(acc.begin(), acc.begin() + (long)members.size()), std::vector<const Ciphertext *>(members.begin(), members.end()));
        }
        for (size_t g = 0; g < groups; g++) ret[0].push_back(std::move(*acc[g]));
        return ret;
    }

    void serializeEncodedWeights(const Plain2d &w, std::ostream &stream) {
        const size_t rows = w.data.size();
        if (rows == 0) throw std::invalid_argument("No rows in weight matrix.");
        const size_t cols = w[0].size();
        if (cols == 0) throw std::invalid_argument("No columns in weight matrix.");
        for (const auto &row : w.data)
            if (row.size() != cols) throw std::invalid_argument("Weight matrix is not rectangular.");
        savet(stream, &rows);
        savet(stream, &cols);
        for (const auto &row : w.data)
            for (const auto &p : row) p.save(stream);
    }
    Plain2d deserializeEncodedWeights(std::istream &stream) {
        size_t rows = 0, cols = 0;
        loadt(stream, &rows);
        loadt(stream, &cols);
        Plain2d ret;
        ret.data.assign(rows, std::vector<Plaintext>(cols));
        for (auto &row : ret.data)
            for (auto &p : row) p.load(stream);
        return ret;
    }

    // only what carries results travels: the result coefficients of c0 (Ciphertext::saveTerms) without packing, whole ciphertexts with it
    void serializeOutputs(troyn::Evaluator &evaluator, const Cipher2d &x, std::ostream &stream) {
        if (!packLwe) {
            tiles([&](size_t di, size_t dj, size_t li, size_t ui, size_t lj, size_t uj) { x[di][dj].saveTerms(stream, evaluator, requiredTerms(li, ui, lj, uj)); });
            return;
        }
        if (x.data[0].size() != packedCount()) throw std::invalid_argument("Output ciphertext count incorrect");
        for (const auto &ct : x[0]) ct.save(stream);
    }
    Cipher2d deserializeOutputs(troyn::Evaluator &evaluator, std::istream &stream) {
        Cipher2d ret;
        if (!packLwe) {
            ret.data.assign(batchBlocks(), std::vector<Ciphertext>(outputBlocks()));
            tiles([&](size_t di, size_t dj, size_t li, size_t ui, size_t lj, size_t uj) { ret[di][dj].loadTerms(stream, evaluator, requiredTerms(li, ui, lj, uj)); });
            return ret;
        }
        ret.data.assign(1, std::vector<Ciphertext>(packedCount()));
        for (auto &ct : ret[0]) ct.load(stream, evaluator.context());
        return ret;
    }
};

// y = conv2d(x, W) mod t, valid padding, stride 1: x [batchSize][inputChannels][imageHeight][imageWidth], W [outputChannels]
// [inputChannels][kernelHeight][kernelWidth], y [batchSize][outputChannels][imageHeight - kernelHeight + 1][imageWidth - kernelWidth + 1]
// (app/LinearHelper.cuh:753-1192).  An image that does not fit a polynomial is cut into overlapping blocks, which become extra batch rows.
class Conv2dHelper {
    using Plaintext = troyn::Plaintext;
    using Ciphertext = troyn::Ciphertext;

    size_t batchSize;
    size_t blockHeight, blockWidth, kernelHeight, kernelWidth;
    size_t imageHeight, imageWidth;
    size_t inputChannels, outputChannels;
    size_t blockBatch, blockInputChannels, blockOutputChannels;
    size_t slotCount;
    int objective;

    size_t outHeight() const { return imageHeight - kernelHeight + 1; }
    size_t outWidth() const { return imageWidth - kernelWidth + 1; }
    size_t validHeight() const { return blockHeight - kernelHeight + 1; } // results one block yields per side
    size_t validWidth() const { return blockWidth - kernelWidth + 1; }
    size_t cutsDown() const { return ceilDiv(outHeight(), validHeight()); }
    size_t cutsAcross() const { return ceilDiv(outWidth(), validWidth()); }
    size_t blockSize() const { return blockHeight * blockWidth; }
    size_t inputGroups() const { return ceilDiv(inputChannels, blockInputChannels); }
    size_t outputGroups() const { return ceilDiv(outputChannels, blockOutputChannels); }
    // coefficient of a result polynomial holding (batch row b, output channel c of the group, valid pixel (i, j)) of a block
    size_t resultCoeff(size_t b, size_t c, size_t i, size_t j) const {
        return ((b * blockOutputChannels + c) * blockInputChannels + blockInputChannels - 1) * blockSize() + (blockHeight - validHeight() + i) * blockWidth +
               (blockWidth - validWidth() + j);
    }
    std::vector<size_t> requiredTerms() const {
        std::vector<size_t> terms;
        terms.reserve(validHeight() * validWidth() * blockBatch * blockOutputChannels);
        for (size_t b = 0; b < blockBatch; b++)
            for (size_t c = 0; c < blockOutputChannels; c++)
                for (size_t i = 0; i < validHeight(); i++)
                    for (size_t j = 0; j < validWidth(); j++) terms.push_back(resultCoeff(b, c, i, j));
        return terms;
    }
    // visit every result coefficient of result ciphertext (eb, group lc / blockOutputChannels): f(coefficient, index into y)
    template <class F> void results(size_t eb, size_t lc, F f) const {
        const size_t cuts = cutsDown() * cutsAcross(), si = (eb % cuts) / cutsAcross(), sj = eb % cutsAcross();
        const size_t lb = (eb / cuts) * blockBatch, ub = std::min(lb + blockBatch, batchSize), uc = std::min(lc + blockOutputChannels, outputChannels);
        for (size_t b = lb; b < ub; b++)
            for (size_t c = lc; c < uc; c++)
                for (size_t i = 0; i < validHeight(); i++)
                    for (size_t j = 0; j < validWidth(); j++) {
                        const size_t y = si * validHeight() + i, x = sj * validWidth() + j;
                        if (y < outHeight() && x < outWidth()) f(resultCoeff(b - lb, c - lc, i, j), ((b * outputChannels + c) * outHeight() + y) * outWidth() + x);
                    }
    }


public:
    // blocks (b, h, w, ci, co) with b h w ci co <= N that minimise the ciphertexts that travel (LinearHelper.cuh:779-841): large blocks
    // first, ties keep the first found
    Conv2dHelper(size_t batchSize, size_t imageHeight, size_t imageWidth, size_t kernelHeight, size_t kernelWidth, size_t inputChannels, size_t outputChannels,
                 size_t slotCount, int objective = 0)
        : batchSize(batchSize), kernelHeight(kernelHeight), kernelWidth(kernelWidth), imageHeight(imageHeight), imageWidth(imageWidth),
          inputChannels(inputChannels), outputChannels(outputChannels), slotCount(slotCount), objective(objective) {
        if (objective < 0 || objective > 2) throw std::runtime_error("Conv2dHelper: invalid objective");
        size_t best = 2147483647;
        blockBatch = blockHeight = blockWidth = blockInputChannels = blockOutputChannels = 0;
        for (size_t b = batchSize; b >= 1; b--)
            for (size_t h = std::min(imageHeight, slotCount / b); h >= kernelHeight; h--)
                for (size_t w = std::min(imageWidth, slotCount / b / h); w >= kernelWidth; w--)
                    for (size_t co = std::min(outputChannels, slotCount / b / h / w); co >= 1; co--) {
                        const size_t ci = std::min(slotCount / b / h / w / co, inputChannels);
                        if (ci == 0) continue;
                        const size_t images = ceilDiv(batchSize, b) * ceilDiv(imageHeight - kernelHeight + 1, h - kernelHeight + 1) *
                                              ceilDiv(imageWidth - kernelWidth + 1, w - kernelWidth + 1);
                        const size_t in = images * ceilDiv(inputChannels, ci), out = images * ceilDiv(outputChannels, co);
                        const size_t weight = ceilDiv(inputChannels, ci) * ceilDiv(outputChannels, co);
                        const size_t cost = objective == 0 ? in + out : objective == 1 ? weight + out : out + in + weight;
                        if (cost < best) { best = cost; blockBatch = b; blockHeight = h; blockWidth = w; blockInputChannels = ci; blockOutputChannels = co; }
                    }
    }

    size_t getTotalBatchSize() { return ceilDiv(batchSize, blockBatch) * cutsDown() * cutsAcross(); }

    // weights[oc][ic] flipped, input channel ic of the group at plane (blockInputChannels - 1 - (ic - lic)) of output channel (oc - loc)
    Plain2d encodeWeights(troyn::BatchEncoder &encoder, std::vector<uint64_t> weights) {
        if (weights.size() != inputChannels * outputChannels * kernelHeight * kernelWidth) throw std::invalid_argument("Weights shape incorrect.");
        Plain2d encoded;
        encoded.data.assign(outputGroups(), std::vector<Plaintext>(inputGroups()));
        std::vector<uint64_t> poly(blockInputChannels * blockOutputChannels * blockSize());
        for (size_t go = 0; go < outputGroups(); go++)
            for (size_t gi = 0; gi < inputGroups(); gi++) {
                std::fill(poly.begin(), poly.end(), 0);
                const size_t loc = go * blockOutputChannels, lic = gi * blockInputChannels;
                for (size_t oc = loc; oc < std::min(loc + blockOutputChannels, outputChannels); oc++)
                    for (size_t ic = lic; ic < std::min(lic + blockInputChannels, inputChannels); ic++)
                        for (size_t ki = 0; ki < kernelHeight; ki++)
                            for (size_t kj = 0; kj < kernelWidth; kj++)
                                poly[((oc - loc) * blockInputChannels + (blockInputChannels - 1 - (ic - lic))) * blockSize() + ki * blockWidth + kj] =
                                    weights[((oc * inputChannels + ic) * kernelHeight + (kernelHeight - 1 - ki)) * kernelWidth + (kernelWidth - 1 - kj)];
                encoder.encodePolynomial(poly, encoded[go][gi]);
            }
        return encoded;
    }

    // (batch block, vertical cut, horizontal cut) -> one row of ceil(inputChannels / blockInputChannels) polynomials; image b of the block
    // starts at coefficient b * blockInputChannels * blockOutputChannels * blockSize, channel plane tci at tci * blockSize
    Plain2d encodeInputs(troyn::BatchEncoder &encoder, const std::vector<uint64_t> &inputs) {
        if (inputs.size() != batchSize * inputChannels * imageHeight * imageWidth) throw std::invalid_argument("Inputs shape incorrect.");
        const size_t imageSize = imageHeight * imageWidth;
        Plain2d ret;
        ret.data.reserve(getTotalBatchSize());
        std::vector<uint64_t> poly(slotCount);
        for (size_t lb = 0; lb < batchSize; lb += blockBatch)
            for (size_t ih = 0; ih < cutsDown(); ih++)
                for (size_t iw = 0; iw < cutsAcross(); iw++) {
                    const size_t si = ih * validHeight(), sj = iw * validWidth();
                    const size_t ui = std::min(si + blockHeight, imageHeight), uj = std::min(sj + blockWidth, imageWidth), ub = std::min(lb + blockBatch, batchSize);
                    std::vector<Plaintext> group(inputGroups());
                    for (size_t g = 0; g < inputGroups(); g++) {
                        std::fill(poly.begin(), poly.end(), 0);
                        const size_t lci = g * blockInputChannels, uci = std::min(lci + blockInputChannels, inputChannels);
                        for (size_t b = lb; b < ub; b++)
                            for (size_t ci = lci; ci < uci; ci++)
                                for (size_t ti = si; ti < ui; ti++)
                                    for (size_t tj = sj; tj < uj; tj++)
                                        poly[((b - lb) * blockInputChannels * blockOutputChannels + (ci - lci)) * blockSize() + (ti - si) * blockWidth + (tj - sj)] =
                                            inputs[(b * inputChannels + ci) * imageSize + ti * imageWidth + tj];
                        encoder.encodePolynomial(poly, group[g]);
                    }
                    ret.data.push_back(std::move(group));
                }
        return ret;
    }

    Cipher2d encryptInputs(const troyn::Encryptor &encryptor, troyn::BatchEncoder &encoder, const std::vector<uint64_t> &inputs) {
        return encodeInputs(encoder, inputs).encrypt(encryptor);
    }

    // ret[b][oc] = sum_i a[b][i] * weights[oc][i]
    Cipher2d conv2d(const troyn::Evaluator &evaluator, const Cipher2d &a, const Plain2d &encodedWeights) {
        return detail::accumulatePlain(evaluator, getTotalBatchSize(), outputGroups(), a.data.empty() ? 0 : a[0].size(), true,
                                       [&](size_t b, size_t i, size_t) -> const Ciphertext & { return a[b][i]; },
                                       [&](size_t, size_t i, size_t oc) -> const Plaintext & { return encodedWeights[oc][i]; });
    }
    Cipher2d conv2dCipher(const troyn::Evaluator &evaluator, const Cipher2d &a, const Cipher2d &encodedWeights) {
        return detail::accumulateCipher(evaluator, getTotalBatchSize(), outputGroups(), a.data.empty() ? 0 : a[0].size(),
                                        [&](size_t b, size_t i) -> const Ciphertext & { return a[b][i]; }, [&](size_t i, size_t oc) -> const Ciphertext & { return encodedWeights[oc][i]; });
    }
    Cipher2d conv2dReverse(const troyn::Evaluator &evaluator, const Plain2d &a, const Cipher2d &encodedWeights) {
        return detail::accumulatePlain(evaluator, getTotalBatchSize(), outputGroups(), a.data.empty() ? 0 : a[0].size(), false,
                                       [&](size_t, size_t i, size_t oc) -> const Ciphertext & { return encodedWeights[oc][i]; },
                                       [&](size_t b, size_t i, size_t) -> const Plaintext & { return a[b][i]; });
    }

    // a bias / expected output in the layout of conv2d's result.  As in the reference (LinearHelper.cuh:1035-1078) the coefficient
    // buffer is NOT cleared between result ciphertexts: a border block, whose out-of-image pixels are skipped, keeps the values the
    // previous block wrote there -- decryptOutputs never reads them
    Plain2d encodeOutputs(troyn::BatchEncoder &encoder, const std::vector<uint64_t> &outputs) {
        if (outputs.size() != batchSize * outputChannels * outHeight() * outWidth()) throw std::invalid_argument("Outputs shape incorrect.");
        const size_t total = getTotalBatchSize();
        Plain2d ret;
        ret.data.assign(total, std::vector<Plaintext>(outputGroups()));
        std::vector<uint64_t> poly(slotCount, 0);
        for (size_t eb = 0; eb < total; eb++)
            for (size_t g = 0; g < outputGroups(); g++) {
                results(eb, g * blockOutputChannels, [&](size_t coeff, size_t index) { poly[coeff] = outputs[index]; });
                encoder.encodePolynomial(poly, ret[eb][g]);
            }
        return ret;
    }

    std::vector<uint64_t> decryptOutputs(troyn::BatchEncoder &encoder, troyn::Decryptor &decryptor, const Cipher2d &outputs) {
        std::vector<uint64_t> ret(batchSize * outputChannels * outHeight() * outWidth(), 0), coeffs;
        Plaintext pt;
        const size_t total = getTotalBatchSize();
        for (size_t eb = 0; eb < total; eb++)
            for (size_t g = 0; g < outputGroups(); g++) {
                decryptor.decrypt(outputs[eb][g], pt);
                encoder.decodePolynomial(pt, coeffs);
                coeffs.resize(slotCount, 0);
                results(eb, g * blockOutputChannels, [&](size_t coeff, size_t index) { ret[index] = coeffs[coeff]; });
            }
        return ret;
    }

    void serializeOutputs(troyn::Evaluator &evaluator, const Cipher2d &x, std::ostream &stream) {
        const std::vector<size_t> terms = requiredTerms();
        const size_t total = getTotalBatchSize();
        for (size_t b = 0; b < total; b++)
            for (size_t g = 0; g < outputGroups(); g++) x[b][g].saveTerms(stream, evaluator, terms);
    }
    // rows come back `outputChannels` wide with the first ceil(outputChannels / blockOutputChannels) entries filled, as in the reference
    Cipher2d deserializeOutputs(troyn::Evaluator &evaluator, std::istream &stream) {
        const std::vector<size_t> terms = requiredTerms();
        Cipher2d ret;
        ret.data.assign(getTotalBatchSize(), std::vector<Ciphertext>(outputChannels));
        for (auto &row : ret.data)
            for (size_t g = 0; g < outputGroups(); g++) row[g].loadTerms(stream, evaluator, terms);
        return ret;
    }
};

} // namespace LinearHelper
