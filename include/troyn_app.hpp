// troyn_app.hpp -- the reference's CKKS linear-layer helpers (app/LinearHelperCKKS.cuh: Plain2d, Cipher2d, MatmulHelper,
// Conv2dHelper) over the troyn:: mirror in troyn.hpp, i.e. over libtroyhip.so.  Same namespace, class and member names, argument
// order and exceptions, so code written against the reference's app header compiles against this one.
//
// The packing (Cheetah-style coefficient encoding): a plaintext polynomial holds a weight block so that the product
// x(X) * w(X) carries the inner products <x, w[:, j]> at coefficient (j + 1) * blockHeight - 1; nothing here rotates or
// relinearises -- the GPU work is multiplyPlain + add, which is why the helpers matter for the hot path's cfgE.
//
// What is MI355X-specific: the ciphertexts of one Cipher2d column live in ONE device slab (troyn::Ciphertext::allocateBatch), and
// the helpers hand a whole column to the library as a single batched launch (Evaluator::multiplyPlainBatch / addInplaceBatch) when
// they find that layout -- ONE launch per output block (the sum over the input blocks inside it: troyhip_multiply_plain_accumulate) for
// a 128x128 layer at any batch size instead of 3 kernels per input row.  Inputs that were assembled ciphertext by ciphertext take the
// reference's per-element loop.
#pragma once
#include "troyn.hpp"
#include <cassert>
#include <functional>

namespace LinearHelperCKKS {

template <typename T> inline void savet(std::ostream &stream, const T *obj) { stream.write(reinterpret_cast<const char *>(obj), sizeof(T)); }
template <typename T> inline void loadt(std::istream &stream, T *obj) { stream.read(reinterpret_cast<char *>(obj), sizeof(T)); }

inline static size_t ceilDiv(size_t a, size_t b) { return (a + b - 1) / b; }

class Plain2d { // app/LinearHelperCKKS.cuh:17-32
public:
    std::vector<std::vector<troyn::Plaintext>> data;
    std::vector<troyn::Plaintext> &operator[](size_t id) { return data[id]; }
    const std::vector<troyn::Plaintext> &operator[](size_t id) const { return data[id]; }
    Plain2d() {}
};

class Cipher2d { // app/LinearHelperCKKS.cuh:34-96
public:
    std::vector<std::vector<troyn::Ciphertext>> data;
    std::vector<troyn::Ciphertext> &operator[](size_t id) { return data[id]; }
    const std::vector<troyn::Ciphertext> &operator[](size_t id) const { return data[id]; }
    Cipher2d() {}

    // layout: rows, columns (size_t each), then the ciphertexts row-major in Ciphertext::save format; an empty grid writes nothing
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
        read(stream, [&](troyn::Ciphertext &ct) { ct.load(stream); });
    }
    void load(std::istream &stream, const troyn::SEALContext &context) {
        read(stream, [&](troyn::Ciphertext &ct) { ct.load(stream, context); });
    }

private:
    void read(std::istream &stream, const std::function<void(troyn::Ciphertext &)> &one) {
        size_t rows = 0, cols = 0;
        loadt(stream, &rows);
        loadt(stream, &cols);
        data.assign(rows, std::vector<troyn::Ciphertext>());
        for (auto &row : data) {
            row.resize(cols);
            for (auto &ct : row) one(ct);
        }
    }
};

namespace detail {

// element-wise y op= x over two grids of the same shape (the shape check and message of the reference's addPlainInplace / addInplace)
template <class Y, class X, class F> inline void zip(Y &y, const X &x, F f) {
    if (y.data.size() != x.data.size()) throw std::invalid_argument("Size incorrect.");
    for (size_t i = 0; i < y.data.size(); i++) {
        if (y[i].size() != x[i].size()) throw std::invalid_argument("Size incorrect.");
        for (size_t j = 0; j < y[i].size(); j++) f(y[i][j], x[i][j]);
    }
}

// encrypt every plaintext of a grid with the secret key.  The ciphertexts of one COLUMN (same block of every batch row) are
// carved out of one device slab so that the layer's multiply-accumulate can take the column in a single launch.
inline Cipher2d encryptGrid(const troyn::Encryptor &encryptor, const Plain2d &plain) {
    Cipher2d out;
    const size_t rows = plain.data.size();
    out.data.resize(rows);
    if (!rows) return out;
    bool rect = true;
    for (auto &r : plain.data) rect = rect && r.size() == plain.data[0].size();
    if (!rect) { // ragged: no column structure to exploit
        for (size_t i = 0; i < rows; i++)
            for (auto &p : plain[i]) out[i].push_back(encryptor.encryptSymmetric(p));
        return out;
    }
    const size_t cols = plain.data[0].size();
    for (size_t i = 0; i < rows; i++) out[i].resize(cols);
    for (size_t j = 0; j < cols; j++) {
        std::vector<troyn::Ciphertext> fresh(rows);
        for (size_t i = 0; i < rows; i++) encryptor.encryptSymmetric(plain[i][j], fresh[i]);
        std::vector<troyn::Ciphertext> column = troyn::Ciphertext::packBatch(fresh);
        for (size_t i = 0; i < rows; i++) out[i][j] = std::move(column[i]);
    }
    return out;
}

// out[b][o] = sum_i a[b][i] * weight(o, i) for every batch row b: one batched multiplyPlain (+ add) per (o, i) when column i of `a`
// is a slab, the reference's ciphertext-by-ciphertext loop otherwise
template <class W> inline Cipher2d multiplyAccumulate(const troyn::Evaluator &evaluator, const Cipher2d &a, size_t outputs, size_t inputs, W weight) {
    const size_t rows = a.data.size();
    Cipher2d ret;
    ret.data.resize(rows);
    for (auto &r : ret.data) r.resize(outputs);
    bool slabs = rows > 1;
    std::vector<const troyn::Ciphertext *> column(rows);
    for (size_t i = 0; i < inputs && slabs; i++) {
        for (size_t b = 0; b < rows; b++) column[b] = &a[b][i];
        slabs = troyn::Ciphertext::isBatch(column);
    }
    if (!slabs) {
        for (size_t b = 0; b < rows; b++)
            for (size_t o = 0; o < outputs; o++)
                for (size_t i = 0; i < inputs; i++) {
                    troyn::Ciphertext prod;
                    evaluator.multiplyPlain(a[b][i], weight(o, i), prod);
                    if (i == 0) ret[b][o] = std::move(prod);
                    else evaluator.addInplace(ret[b][o], prod);
                }
        return ret;
    }
    for (size_t o = 0; o < outputs; o++) {
        std::vector<troyn::Ciphertext> acc;
        if (inputs <= 16) { // the whole sum over the input blocks in one pass
            std::vector<std::vector<const troyn::Ciphertext *>> columns(inputs, std::vector<const troyn::Ciphertext *>(rows));
            std::vector<const troyn::Plaintext *> plains(inputs);
            for (size_t i = 0; i < inputs; i++) {
                for (size_t b = 0; b < rows; b++) columns[i][b] = &a[b][i];
                plains[i] = &weight(o, i);
            }
            acc = evaluator.multiplyPlainAccumulateBatch(columns, plains);
        } else {
            for (size_t i = 0; i < inputs; i++) {
                for (size_t b = 0; b < rows; b++) column[b] = &a[b][i];
                std::vector<troyn::Ciphertext> prod = evaluator.multiplyPlainBatch(column, weight(o, i));
                if (i == 0) acc = std::move(prod);
                else evaluator.addInplaceBatch(acc, prod);
            }
        }
        for (size_t b = 0; b < rows; b++) ret[b][o] = std::move(acc[b]);
    }
    return ret;
}

} // namespace detail

// y = x W for x [batchSize][inputDims], W [inputDims][outputDims] (row-major doubles), app/LinearHelperCKKS.cuh:104-360
class MatmulHelper {
    using Plaintext = troyn::Plaintext;
    using Ciphertext = troyn::Ciphertext;

    size_t batchSize, inputDims, outputDims;
    size_t slotCount;
    size_t blockHeight, blockWidth;

    // block of blockHeight input dims x blockWidth output dims per weight polynomial (blockHeight * blockWidth <= N), chosen to
    // minimise the number of ciphertexts that travel: ceil(in / h) up + ceil(out / w) down (LinearHelperCKKS.cuh:112-123)
    void determineBlock() {
        const size_t coeffs = slotCount * 2;
        size_t best = inputDims + outputDims + 1;
        blockHeight = blockWidth = 0;
        for (size_t h = 1; h <= inputDims; h++) {
            const size_t w = std::min(coeffs / h, outputDims);
            if (!w) break;
            const size_t cost = ceilDiv(inputDims, h) + ceilDiv(outputDims, w);
            if (cost < best) { best = cost; blockHeight = h; blockWidth = w; }
        }
    }
    size_t inputBlocks() const { return ceilDiv(inputDims, blockHeight); }
    size_t outputBlocks() const { return ceilDiv(outputDims, blockWidth); }
    // coefficient of an output polynomial that holds output dim (first of the block + k)
    size_t outputCoeff(size_t k) const { return (k + 1) * blockHeight - 1; }
    std::vector<size_t> requiredTerms(size_t block) const { // LinearHelperCKKS.cuh:326-337
        const size_t lo = block * blockWidth, hi = std::min(lo + blockWidth, outputDims);
        std::vector<size_t> terms(hi - lo);
        for (size_t k = 0; k < terms.size(); k++) terms[k] = outputCoeff(k);
        return terms;
    }

public:
    Plain2d encodedWeights;

    MatmulHelper(size_t batchSize, size_t inputDims, size_t outputDims, size_t slotCount)
        : batchSize(batchSize), inputDims(inputDims), outputDims(outputDims), slotCount(slotCount) {
        determineBlock();
    }

    // encodedWeights[bi][bj](X) = sum_{i, j in block} W[i][j] X^{(j - lj) * h + (h - 1) - (i - li)}: input dim i of the block meets
    // X^{i - li} of the input polynomial at exponent (j - lj) * h + h - 1
    void encodeWeights(troyn::CKKSEncoder &encoder, troyn::ParmsID parmsID, const std::vector<double> &weights, double scale) {
        if (weights.size() != inputDims * outputDims) throw std::invalid_argument("Weight size incorrect.");
        const size_t h = blockHeight, w = blockWidth;
        encodedWeights.data.assign(inputBlocks(), std::vector<Plaintext>(outputBlocks()));
        std::vector<double> poly(slotCount * 2);
        for (size_t bi = 0; bi < inputBlocks(); bi++)
            for (size_t bj = 0; bj < outputBlocks(); bj++) {
                std::fill(poly.begin(), poly.end(), 0.0);
                for (size_t i = bi * h; i < std::min((bi + 1) * h, inputDims); i++)
                    for (size_t j = bj * w; j < std::min((bj + 1) * w, outputDims); j++)
                        poly[(j - bj * w) * h + (h - 1) - (i - bi * h)] = weights[i * outputDims + j];
                encoder.encodePolynomial(poly, parmsID, scale, encodedWeights[bi][bj]);
            }
    }

    // row b -> ceil(inputDims / h) polynomials whose coefficients are consecutive input dims
    Plain2d encodeInputs(troyn::CKKSEncoder &encoder, troyn::ParmsID parmsID, const std::vector<double> &inputs, double scale) {
        if (inputs.size() != inputDims * batchSize) throw std::invalid_argument("Input size incorrect.");
        Plain2d ret;
        ret.data.assign(batchSize, std::vector<Plaintext>(inputBlocks()));
        for (size_t b = 0; b < batchSize; b++)
            for (size_t bi = 0; bi < inputBlocks(); bi++) {
                const double *first = inputs.data() + b * inputDims + bi * blockHeight;
                const size_t count = std::min(blockHeight, inputDims - bi * blockHeight);
                encoder.encodePolynomial(std::vector<double>(first, first + count), parmsID, scale, ret[b][bi]);
            }
        return ret;
    }

    Cipher2d encryptInputs(const troyn::Encryptor &encryptor, troyn::CKKSEncoder &encoder, troyn::ParmsID parmsID, const std::vector<double> &inputs, double scale) {
        return detail::encryptGrid(encryptor, encodeInputs(encoder, parmsID, inputs, scale));
    }

    // ret[b][bj] = sum_bi a[b][bi] * encodedWeights[bi][bj]
    Cipher2d matmul(const troyn::Evaluator &evaluator, const Cipher2d &a) {
        if (a.data.size() != batchSize) throw std::invalid_argument("Input batchsize incorrect.");
        for (const auto &row : a.data)
            if (row.size() != encodedWeights.data.size()) throw std::invalid_argument("Input batchsize incorrect.");
        return detail::multiplyAccumulate(evaluator, a, outputBlocks(), encodedWeights.data.size(),
                                          [&](size_t o, size_t i) -> const Plaintext & { return encodedWeights[i][o]; });
    }

    // a bias / expected output laid out as matmul's result (only the coefficients decryptOutputs reads are set)
    Plain2d encodeOutputs(troyn::CKKSEncoder &encoder, troyn::ParmsID parmsID, const std::vector<double> &outputs, double scale) {
        if (outputs.size() != batchSize * outputDims) throw std::invalid_argument("Output size incorrect.");
        Plain2d ret;
        ret.data.assign(batchSize, std::vector<Plaintext>(outputBlocks()));
        std::vector<double> poly(slotCount * 2);
        for (size_t b = 0; b < batchSize; b++)
            for (size_t bj = 0; bj < outputBlocks(); bj++) {
                std::fill(poly.begin(), poly.end(), 0.0);
                const size_t lo = bj * blockWidth, hi = std::min(lo + blockWidth, outputDims);
                for (size_t j = lo; j < hi; j++) poly[outputCoeff(j - lo)] = outputs[b * outputDims + j];
                encoder.encodePolynomial(poly, parmsID, scale, ret[b][bj]);
            }
        return ret;
    }

    void addPlainInplace(const troyn::Evaluator &evaluator, Cipher2d &y, const Plain2d &x) {
        detail::zip(y, x, [&](Ciphertext &c, const Plaintext &p) { evaluator.addPlainInplace(c, p); });
    }

    std::vector<double> decryptOutputs(troyn::CKKSEncoder &encoder, troyn::Decryptor &decryptor, const Cipher2d &outputs) {
        std::vector<double> dec(batchSize * outputDims), coeffs;
        Plaintext pt;
        for (size_t b = 0; b < batchSize; b++)
            for (size_t bj = 0; bj < outputBlocks(); bj++) {
                decryptor.decrypt(outputs[b][bj], pt);
                encoder.decodePolynomial(pt, coeffs);
                const size_t lo = bj * blockWidth, hi = std::min(lo + blockWidth, outputDims);
                for (size_t j = lo; j < hi; j++) dec[b * outputDims + j] = coeffs[outputCoeff(j - lo)];
            }
        return dec;
    }

    // only the coefficients that carry results travel (Ciphertext::saveTerms), LinearHelperCKKS.cuh:322-358
    void serializeOutputs(troyn::Evaluator &evaluator, const Cipher2d &x, std::ostream &stream) {
        for (size_t b = 0; b < batchSize; b++)
            for (size_t bj = 0; bj < outputBlocks(); bj++) x[b][bj].saveTerms(stream, evaluator, requiredTerms(bj));
    }
    Cipher2d deserializeOutputs(troyn::Evaluator &evaluator, std::istream &stream) {
        Cipher2d ret;
        ret.data.assign(batchSize, std::vector<Ciphertext>(outputBlocks()));
        for (size_t b = 0; b < batchSize; b++)
            for (size_t bj = 0; bj < outputBlocks(); bj++) ret[b][bj].loadTerms(stream, evaluator, requiredTerms(bj));
        return ret;
    }
};

// y = conv2d(x, W), valid padding, stride 1: x [batchSize][inputChannels][imageHeight][imageWidth],
// W [outputChannels][inputChannels][kernelHeight][kernelWidth], y [batchSize][outputChannels][imageHeight - kernelHeight + 1]
// [imageWidth - kernelWidth + 1] (app/LinearHelperCKKS.cuh:362-713).  Images larger than sqrt(N) per side are cut into
// overlapping blocks that become extra batch rows.
class Conv2dHelper {
    using Plaintext = troyn::Plaintext;
    using Ciphertext = troyn::Ciphertext;

    size_t batchSize;
    size_t blockHeight, blockWidth, kernelHeight, kernelWidth;
    size_t imageHeight, imageWidth;
    size_t inputChannels, outputChannels;
    size_t slotCount;
    bool blocked;

    size_t blockSize() const { return blockHeight * blockWidth; }
    size_t channelSlots() const { return (slotCount * 2) / blockSize(); } // channels that share one polynomial
    size_t channelGroups() const { return ceilDiv(inputChannels, channelSlots()); }
    size_t outHeight() const { return imageHeight - kernelHeight + 1; }
    size_t outWidth() const { return imageWidth - kernelWidth + 1; }
    // how the image is tiled: tiles advance by (block - kernel + 1) so every output pixel is complete in exactly one tile
    size_t stepH() const { return blockHeight - kernelHeight + 1; }
    size_t stepW() const { return blockWidth - kernelWidth + 1; }
    size_t tilesH() const { return ceilDiv(imageHeight - (kernelHeight - 1), stepH()); }
    size_t tilesW() const { return ceilDiv(imageWidth - (kernelWidth - 1), stepW()); }

    // visits (coefficient of the output polynomial, index into the [batch][oc][oh][ow] tensor) of every output pixel that tile row b
    // and output channel c own
    template <class F> void forOutputs(size_t b, size_t c, F f) const {
        const size_t per = tilesH() * tilesW();
        const size_t image = b / per, ti = (b % per) / tilesW(), tj = b % tilesW();
        const size_t base = (channelSlots() - 1) * blockSize();
        for (size_t i = 0; i < stepH(); i++)
            for (size_t j = 0; j < stepW(); j++) {
                const size_t oi = ti * stepH() + i, oj = tj * stepW() + j;
                if (oi >= outHeight() || oj >= outWidth()) continue;
                const size_t coeff = base + (kernelHeight - 1 + i) * blockWidth + (kernelWidth - 1 + j);
                f(coeff, ((image * outputChannels + c) * outHeight() + oi) * outWidth() + oj);
            }
    }
    std::vector<size_t> requiredTerms() const { // the last channel slot of the polynomial (LinearHelperCKKS.cuh:684-690)
        std::vector<size_t> terms(blockSize());
        for (size_t k = 0; k < terms.size(); k++) terms[k] = (channelSlots() - 1) * blockSize() + k;
        return terms;
    }

public:
    Plain2d encodedWeights;

    Conv2dHelper(size_t batchSize, size_t imageHeight, size_t imageWidth, size_t kernelHeight, size_t kernelWidth, size_t inputChannels, size_t outputChannels,
                 size_t slotCount)
        : batchSize(batchSize), kernelHeight(kernelHeight), kernelWidth(kernelWidth), imageHeight(imageHeight), imageWidth(imageWidth),
          inputChannels(inputChannels), outputChannels(outputChannels), slotCount(slotCount) {
        const size_t side = (size_t)std::sqrt((double)(slotCount * 2));
        blocked = imageHeight > side || imageWidth > side;
        blockHeight = blocked ? side : imageHeight;
        blockWidth = blocked ? side : imageWidth;
    }

    // encodedWeights[oc][g]: channel k of group g sits in slot (channelSlots - 1 - k) with its kernel flipped in both axes, so that the
    // product with an input polynomial (channel k in slot k) sums all channels' correlations into the LAST slot
    void encodeWeights(troyn::CKKSEncoder &encoder, troyn::ParmsID parmsID, std::vector<double> weights, double scale) {
        if (weights.size() != inputChannels * outputChannels * kernelHeight * kernelWidth) throw std::invalid_argument("Weights shape incorrect.");
        const size_t cs = channelSlots(), ksz = kernelHeight * kernelWidth;
        encodedWeights.data.assign(outputChannels, std::vector<Plaintext>(channelGroups()));
        std::vector<double> poly(cs * blockSize());
        for (size_t oc = 0; oc < outputChannels; oc++)
            for (size_t g = 0; g < channelGroups(); g++) {
                std::fill(poly.begin(), poly.end(), 0.0);
                for (size_t ic = g * cs; ic < std::min((g + 1) * cs, inputChannels); ic++) {
                    const double *kernel = weights.data() + (oc * inputChannels + ic) * ksz;
                    double *slot = poly.data() + (cs - 1 - (ic - g * cs)) * blockSize();
                    for (size_t u = 0; u < kernelHeight; u++)
                        for (size_t v = 0; v < kernelWidth; v++) slot[u * blockWidth + v] = kernel[(kernelHeight - 1 - u) * kernelWidth + (kernelWidth - 1 - v)];
                }
                encoder.encodePolynomial(poly, parmsID, scale, encodedWeights[oc][g]);
            }
    }

    size_t getTotalBatchSize() { return blocked ? batchSize * tilesH() * tilesW() : batchSize; }

    Plain2d encodeInputs(troyn::CKKSEncoder &encoder, troyn::ParmsID parmsID, const std::vector<double> &inputs, double scale) {
        if (inputs.size() != batchSize * inputChannels * imageHeight * imageWidth) throw std::invalid_argument("Inputs shape incorrect.");
        const size_t total = getTotalBatchSize(), cs = channelSlots(), per = blocked ? tilesH() * tilesW() : 1;
        Plain2d ret;
        ret.data.assign(total, std::vector<Plaintext>(channelGroups()));
        std::vector<double> poly(slotCount * 2);
        for (size_t b = 0; b < total; b++) {
            // tile origin in the image (the whole image when not blocked)
            const size_t image = b / per, oi = blocked ? ((b % per) / tilesW()) * stepH() : 0, oj = blocked ? (b % tilesW()) * stepW() : 0;
            const size_t rows = std::min(blockHeight, imageHeight - oi), cols = std::min(blockWidth, imageWidth - oj);
            for (size_t g = 0; g < channelGroups(); g++) {
                std::fill(poly.begin(), poly.end(), 0.0);
                for (size_t ic = g * cs; ic < std::min((g + 1) * cs, inputChannels); ic++) {
                    const double *plane = inputs.data() + (image * inputChannels + ic) * imageHeight * imageWidth;
                    double *slot = poly.data() + (ic - g * cs) * blockSize();
                    for (size_t i = 0; i < rows; i++)
                        for (size_t j = 0; j < cols; j++) slot[i * blockWidth + j] = plane[(oi + i) * imageWidth + (oj + j)];
                }
                encoder.encodePolynomial(poly, parmsID, scale, ret[b][g]);
            }
        }
        return ret;
    }

    Cipher2d encryptInputs(const troyn::Encryptor &encryptor, troyn::CKKSEncoder &encoder, troyn::ParmsID parmsID, const std::vector<double> &inputs, double scale) {
        return detail::encryptGrid(encryptor, encodeInputs(encoder, parmsID, inputs, scale));
    }

    // ret[b][oc] = sum_g a[b][g] * encodedWeights[oc][g]
    Cipher2d conv2d(const troyn::Evaluator &evaluator, const Cipher2d &a) {
        if (a.data.size() != getTotalBatchSize()) throw std::invalid_argument("Input batchsize incorrect.");
        const size_t groups = a.data.empty() ? 0 : a.data[0].size();
        return detail::multiplyAccumulate(evaluator, a, outputChannels, groups, [&](size_t o, size_t i) -> const Plaintext & { return encodedWeights[o][i]; });
    }

    Plain2d encodeOutputs(troyn::CKKSEncoder &encoder, troyn::ParmsID parmsID, const std::vector<double> &outputs, double scale) {
        if (outputs.size() != batchSize * outputChannels * outHeight() * outWidth()) throw std::invalid_argument("Outputs shape incorrect.");
        const size_t total = getTotalBatchSize();
        Plain2d ret;
        ret.data.assign(total, std::vector<Plaintext>(outputChannels));
        // the reference reuses ONE coefficient buffer across all (tile, channel) pairs without clearing it
        // (LinearHelperCKKS.cuh:569-596), so positions a later tile does not own keep the previous tile's values; kept as is
        std::vector<double> poly(channelSlots() * blockSize(), 0.0);
        for (size_t b = 0; b < total; b++)
            for (size_t c = 0; c < outputChannels; c++) {
                forOutputs(b, c, [&](size_t coeff, size_t index) { poly[coeff] = outputs[index]; });
                encoder.encodePolynomial(poly, parmsID, scale, ret[b][c]);
            }
        return ret;
    }

    void addPlainInplace(const troyn::Evaluator &evaluator, Cipher2d &y, const Plain2d &x) {
        detail::zip(y, x, [&](Ciphertext &c, const Plaintext &p) { evaluator.addPlainInplace(c, p); });
    }
    void addInplace(const troyn::Evaluator &evaluator, Cipher2d &y, const Cipher2d &x) {
        detail::zip(y, x, [&](Ciphertext &c, const Ciphertext &d) { evaluator.addInplace(c, d); });
    }

    std::vector<double> decryptOutputs(troyn::CKKSEncoder &encoder, troyn::Decryptor &decryptor, const Cipher2d &outputs) {
        std::vector<double> ret(batchSize * outputChannels * outHeight() * outWidth()), coeffs;
        Plaintext pt;
        const size_t total = getTotalBatchSize();
        for (size_t b = 0; b < total; b++)
            for (size_t c = 0; c < outputChannels; c++) {
                decryptor.decrypt(outputs[b][c], pt);
                encoder.decodePolynomial(pt, coeffs);
                forOutputs(b, c, [&](size_t coeff, size_t index) { ret[index] = coeffs[coeff]; });
            }
        return ret;
    }

    void serializeOutputs(troyn::Evaluator &evaluator, const Cipher2d &x, std::ostream &stream) {
        const std::vector<size_t> terms = requiredTerms();
        const size_t total = getTotalBatchSize();
        for (size_t b = 0; b < total; b++)
            for (size_t oc = 0; oc < outputChannels; oc++) x[b][oc].saveTerms(stream, evaluator, terms);
    }
    Cipher2d deserializeOutputs(troyn::Evaluator &evaluator, std::istream &stream) {
        const std::vector<size_t> terms = requiredTerms();
        Cipher2d ret;
        ret.data.assign(getTotalBatchSize(), std::vector<Ciphertext>(outputChannels));
        for (auto &row : ret.data)
            for (auto &ct : row) ct.loadTerms(stream, evaluator, terms);
        return ret;
    }
};

} // namespace LinearHelperCKKS
