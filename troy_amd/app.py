"""Host-side application helpers on top of the evaluator: the reference's `app/LinearHelperCKKS.cuh`
MatmulHelper and the CKKS polynomial (coefficient) encoding it uses, `CKKSEncoderCuda::encodePolynomial / decodePolynomial`
(`src/ckks_cuda.cu:455-575, 983-1055`).  Pure packing logic: every ciphertext operation goes through `troy_amd.api`.

The helper's batch dimension (independent input rows) IS the evaluator's batch dimension here: `Cipher2d[i]` is ONE batched
ciphertext holding block i of every input row, so a matmul over B rows costs the same launches as over one.
`encodePolynomial` exists only in the reference's CUDA encoder (no CPU twin to pin against): it is restated from the CUDA
source and checked by round trip and by the plaintext matmul (floating point, tolerance in the tests).
"""
import math

import numpy as np

from . import api, capi


def _c_round(x):
    """C round(): half away from zero (numpy rounds half to even)."""
    return np.sign(x) * np.floor(np.abs(x) + 0.5)


class CKKSPolyEncoder:
    def __init__(self, context):
        if context.scheme != capi.CKKS:
            raise capi.InvalidArgument(capi.INVALID_ARGUMENT, "unsupported scheme")
        self.context = context
        self.slots = context.N // 2

    def encodePolynomial(self, values, limbs, scale):
        """values: up to N doubles (coefficients) -> uint64 [limbs][N], NTT form, plaintext scale = `scale`."""
        ctx, N = self.context, self.context.N
        v = np.zeros(N, dtype=np.float64)
        values = np.asarray(values, dtype=np.float64)
        if values.size > N:
            raise capi.InvalidArgument(capi.INVALID_ARGUMENT, "values_size is too large")
        v[: values.size] = values
        c = _c_round(v * scale)
        primes = ctx.coeff_modulus[:limbs]
        max_coeff = float(np.max(np.abs(v * scale))) if N else 0.0
        bits = int(math.ceil(math.log2(max(max_coeff, 1.0)))) + 1
        if bits >= sum(int(p).bit_length() for p in primes):
            raise capi.InvalidArgument(capi.INVALID_ARGUMENT, "encoded values are too large")
        out = np.zeros((limbs, N), dtype=np.uint64)
        neg = c < 0
        if bits <= 64:
            mag = np.abs(c).astype(np.uint64)
            for l, p in enumerate(primes):
                r = mag % np.uint64(p)
                out[l] = np.where(neg & (r != 0), np.uint64(p) - r, r)
        else:  # the exact integer value of the (rounded) double, reduced per prime
            ints = [int(x) for x in np.abs(c)]
            for l, p in enumerate(primes):
                r = np.array([x % int(p) for x in ints], dtype=np.uint64)
                out[l] = np.where(neg & (r != 0), np.uint64(p) - r, r)
        buf = api.DeviceBuffer.from_numpy(out)
        ctx.ntt(buf, limbs, primes)
        return buf.to_numpy().reshape(limbs, N)

    def decodePolynomial(self, plain_ntt, scale):
        """uint64 [limbs][N] (NTT form) -> N doubles: inverse NTT, CRT composition, centred, times 1/scale."""
        ctx, N = self.context, self.context.N
        plain_ntt = np.ascontiguousarray(plain_ntt, dtype=np.uint64)
        limbs = plain_ntt.shape[0]
        primes = [int(p) for p in ctx.coeff_modulus[:limbs]]
        buf = api.DeviceBuffer.from_numpy(plain_ntt)
        ctx.ntt(buf, limbs, primes, inverse=True)
        x = buf.to_numpy().reshape(limbs, N)
        q = 1
        for p in primes:
            q *= p
        # CRT: sum_l x_l * (q/p_l) * ((q/p_l)^-1 mod p_l) mod q
        acc = [0] * N
        for l, p in enumerate(primes):
            m = q // p
            w = m * pow(m % p, -1, p)
            xl = x[l]
            for j in range(N):
                acc[j] += int(xl[j]) * w
        half = (q + 1) >> 1
        inv = 1.0 / scale
        out = np.empty(N, dtype=np.float64)
        for j in range(N):
            a = acc[j] % q
            out[j] = float(a - q if a >= half else a) * inv
        return out


def _encrypt(encryptor, plain):
    """the reference's helpers encrypt with the SECRET key (encryptSymmetric, LinearHelperCKKS.cuh:213-222); an Encryptor that only holds a
    public key falls back to it"""
    return encryptor.encryptSymmetric(plain) if getattr(encryptor, "sk", None) is not None else encryptor.encrypt(plain)


def _ceil_div(a, b):
    return (a + b - 1) // b


def _serialize_terms(evaluator, outputs, required_of, stream):
    """stream order of the reference (batch outer, output block / channel inner); each batched ciphertext is brought to
    coefficient form ONCE"""
    hosts = [ct.coeff_host(evaluator) for ct in outputs]
    for b in range(outputs[0].batch):
        for cid, ct in enumerate(outputs):
            ct.saveTerms(stream, evaluator, required_of(cid), index=b, coeff_host=hosts[cid])


def _deserialize_terms(evaluator, context, batch, count, required_of, stream):
    parts = [[None] * batch for _ in range(count)]
    meta = [None] * count
    for b in range(batch):
        for cid in range(count):
            data, ntt, scale, cf = api.Ciphertext.load_terms_host(context, stream, required_of(cid))
            parts[cid][b], meta[cid] = data[0], (ntt, scale, cf)
    outs = []
    for cid in range(count):
        ntt, scale, cf = meta[cid]
        ct = api.Ciphertext.from_numpy(context, np.stack(parts[cid]), False, scale, cf)
        if ntt:
            evaluator.transformToNttInplace(ct)
        outs.append(ct)
    return outs


class MatmulHelper:
    """app/LinearHelperCKKS.cuh:104-360, same packing: an input block of `blockHeight` entries is a polynomial x_0 + x_1 X + ..,
    a weight block (h x w) puts W[i][j] at degree j*h + h-1-i, so that coefficient (j+1)*h - 1 of the product is sum_i x_i W[i][j]."""

    def __init__(self, batchSize, inputDims, outputDims, slotCount):
        self.batchSize, self.inputDims, self.outputDims, self.slotCount = batchSize, inputDims, outputDims, slotCount
        self._determine_block()
        self.encodedWeights = None

    def _determine_block(self):  # LinearHelperCKKS.cuh:112-123
        height, width, slots = self.inputDims, self.outputDims, self.slotCount * 2
        self.blockHeight = self.blockWidth = 0
        bt = height + width + 1
        for i in range(1, height + 1):
            w = min(slots // i, width)
            if w == 0:
                break
            t = _ceil_div(height, i) + _ceil_div(width, w)
            if t < bt:
                self.blockHeight, self.blockWidth, bt = i, w, t

    def encodeWeights(self, encoder, limbs, weights, scale):
        """weights: [inputDims][outputDims] doubles -> Plain2d (list of rows of DeviceBuffer [limbs][N], NTT form)"""
        W = np.asarray(weights, dtype=np.float64).reshape(self.inputDims, self.outputDims)
        h, w, slots = self.blockHeight, self.blockWidth, self.slotCount * 2
        rows = []
        for li in range(0, self.inputDims, h):
            ui = min(li + h, self.inputDims)
            row = []
            for lj in range(0, self.outputDims, w):
                uj = min(lj + w, self.outputDims)
                vec = np.zeros(slots)
                for j in range(lj, uj):
                    for i in range(li, ui):
                        vec[(j - lj) * h + h - (i - li) - 1] = W[i, j]
                row.append(api.DeviceBuffer.from_numpy(encoder.encodePolynomial(vec, limbs, scale)))
            rows.append(row)
        self.encodedWeights, self.weightScale = rows, scale
        return rows

    def encryptInputs(self, encryptor, encoder, limbs, inputs, scale):
        """inputs: [batchSize][inputDims] doubles -> Cipher2d: list over input blocks of ONE batched ciphertext each"""
        X = np.asarray(inputs, dtype=np.float64).reshape(self.batchSize, self.inputDims)
        ctx = encoder.context
        out = []
        for lj in range(0, self.inputDims, self.blockHeight):
            uj = min(lj + self.blockHeight, self.inputDims)
            cts = np.stack([_encrypt(encryptor, encoder.encodePolynomial(X[b, lj:uj], limbs, scale)) for b in range(self.batchSize)])
            out.append(api.Ciphertext.from_numpy(ctx, cts, True, scale, 1))
        return out

    def matmul(self, evaluator, a):
        """Cipher2d x encoded weights -> list over output blocks of one batched ciphertext (same order of additions as the
        reference: block row i = 0 initialises, the others are added in order)"""
        if len(a) != len(self.encodedWeights):
            raise capi.InvalidArgument(capi.INVALID_ARGUMENT, "Input size incorrect.")
        rows, cols = len(self.encodedWeights), len(self.encodedWeights[0])
        if rows <= 16:  # one pass per output block: sum_i a[i] (x) W[i][j], every operand read once (same residues as the loop below)
            return [evaluator.multiplyPlainAccumulate([a[i] for i in range(rows)], [self.encodedWeights[i][j] for i in range(rows)], self.weightScale) for j in range(cols)]
        outs = [None] * cols
        for i, wrow in enumerate(self.encodedWeights):
            for j, wp in enumerate(wrow):
                prod = a[i].copy()
                evaluator.multiplyPlainInplace(prod, wp, self.weightScale)
                if i == 0:
                    outs[j] = prod
                else:
                    evaluator.addInplace(outs[j], prod)
        return outs

    def decryptOutputs(self, evaluator, encoder, secret_key_dev, outputs):
        """-> [batchSize][outputDims] doubles (device decryption, host decode)"""
        dec = np.zeros((self.batchSize, self.outputDims))
        interval, vecsize = self.blockHeight, self.blockWidth
        for cid, li in enumerate(range(0, self.outputDims, vecsize)):
            ui = min(li + vecsize, self.outputDims)
            pts = evaluator.decrypt(outputs[cid], secret_key_dev)  # [batch][limbs][N]
            for b in range(self.batchSize):
                buf = encoder.decodePolynomial(pts[b], outputs[cid].scale)
                for j in range(li, ui):
                    dec[b, j] = buf[(j - li + 1) * interval - 1]
        return dec

    def _required(self, cid):  # LinearHelperCKKS.cuh:326-337: coefficient (j+1)*blockHeight - 1 of output block cid
        li = cid * self.blockWidth
        ui = min(li + self.blockWidth, self.outputDims)
        return [(j - li + 1) * self.blockHeight - 1 for j in range(li, ui)]

    def serializeOutputs(self, evaluator, outputs, stream):
        """LinearHelperCKKS.cuh:326-339: only the coefficients decryptOutputs reads travel (saveTerms)"""
        _serialize_terms(evaluator, outputs, self._required, stream)

    def deserializeOutputs(self, evaluator, context, stream):  # LinearHelperCKKS.cuh:341-358
        return _deserialize_terms(evaluator, context, self.batchSize, _ceil_div(self.outputDims, self.blockWidth), self._required, stream)


class Conv2dHelper:
    """app/LinearHelperCKKS.cuh:362-716: valid (no padding, stride 1) 2-D convolution by polynomial multiplication.  An image block
    (h x w) of channel k sits at degrees k*h*w + i*w + j; the flipped kernel of input channel j at slot (channelSlots-1-j);
    the output pixel (i, j) of the block is read at degree (channelSlots-1)*h*w + (h-yh+i)*w + (w-yw+j).  Images larger than
    sqrt(N) per side are cut into overlapping blocks, which multiplies the batch (`getTotalBatchSize`)."""

    def __init__(self, batchSize, imageHeight, imageWidth, kernelHeight, kernelWidth, inputChannels, outputChannels, slotCount):
        self.batchSize, self.imageHeight, self.imageWidth = batchSize, imageHeight, imageWidth
        self.kernelHeight, self.kernelWidth = kernelHeight, kernelWidth
        self.inputChannels, self.outputChannels, self.slotCount = inputChannels, outputChannels, slotCount
        max_size = int(math.isqrt(slotCount * 2))
        if imageHeight > max_size or imageWidth > max_size:
            self.blockHeight = self.blockWidth = max_size
            self.blocked = True
        else:
            self.blockHeight, self.blockWidth, self.blocked = imageHeight, imageWidth, False
        self.encodedWeights = None

    def _grid(self):
        kh, kw = self.kernelHeight - 1, self.kernelWidth - 1
        return _ceil_div(self.imageHeight - kh, self.blockHeight - kh), _ceil_div(self.imageWidth - kw, self.blockWidth - kw)

    def getTotalBatchSize(self):
        if not self.blocked:
            return self.batchSize
        sh, sw = self._grid()
        return self.batchSize * sh * sw

    def encodeWeights(self, encoder, limbs, weights, scale):
        """weights [outputChannels][inputChannels][kh][kw] -> Plain2d [oc][input-channel group] of DeviceBuffer"""
        W = np.asarray(weights, dtype=np.float64).reshape(self.outputChannels, self.inputChannels, self.kernelHeight, self.kernelWidth)
        bs = self.blockHeight * self.blockWidth
        cs = (self.slotCount * 2) // bs
        rows = []
        for oc in range(self.outputChannels):
            row = []
            for lic in range(0, self.inputChannels, cs):
                uic = min(lic + cs, self.inputChannels)
                spread = np.zeros((cs, self.blockHeight, self.blockWidth))
                for j in range(lic, uic):
                    spread[cs - 1 - (j - lic), : self.kernelHeight, : self.kernelWidth] = W[oc, j, ::-1, ::-1]
                row.append(api.DeviceBuffer.from_numpy(encoder.encodePolynomial(spread.reshape(-1), limbs, scale)))
            rows.append(row)
        self.encodedWeights, self.weightScale = rows, scale
        return rows

    def _split(self, X):
        if not self.blocked:
            return X
        kh, kw = self.kernelHeight - 1, self.kernelWidth - 1
        sh, sw = self._grid()
        out = np.zeros((self.batchSize * sh * sw, self.inputChannels, self.blockHeight, self.blockWidth))
        for b in range(self.batchSize):
            for i in range(sh):
                for j in range(sw):
                    si, sj = i * (self.blockHeight - kh), j * (self.blockWidth - kw)
                    ui, uj = min(si + self.blockHeight, self.imageHeight), min(sj + self.blockWidth, self.imageWidth)
                    out[b * sh * sw + i * sw + j, :, : ui - si, : uj - sj] = X[b, :, si:ui, sj:uj]
        return out

    def encryptInputs(self, encryptor, encoder, limbs, inputs, scale):
        """inputs [batchSize][inputChannels][H][W] -> list over input-channel groups of ONE batched ciphertext (batch = total batch)"""
        X = self._split(np.asarray(inputs, dtype=np.float64).reshape(self.batchSize, self.inputChannels, self.imageHeight, self.imageWidth))
        total, interval = X.shape[0], self.blockHeight * self.blockWidth
        cs = (self.slotCount * 2) // interval
        ctx = encoder.context
        out = []
        for c0 in range(0, self.inputChannels, cs):
            c1 = min(c0 + cs, self.inputChannels)
            cts = np.stack([_encrypt(encryptor, encoder.encodePolynomial(X[b, c0:c1].reshape(-1), limbs, scale)) for b in range(total)])
            out.append(api.Ciphertext.from_numpy(ctx, cts, True, scale, 1))
        return out

    def conv2d(self, evaluator, a):
        """-> list over output channels of one batched ciphertext (sum over input-channel groups in order)"""
        if len(a) <= 16:  # one pass per output channel (troyhip_multiply_plain_accumulate), as in MatmulHelper.matmul
            return [evaluator.multiplyPlainAccumulate(list(a), [self.encodedWeights[oc][i] for i in range(len(a))], self.weightScale) for oc in range(self.outputChannels)]
        outs = []
        for oc in range(self.outputChannels):
            acc = None
            for i, ai in enumerate(a):
                prod = ai.copy()
                evaluator.multiplyPlainInplace(prod, self.encodedWeights[oc][i], self.weightScale)
                if acc is None:
                    acc = prod
                else:
                    evaluator.addInplace(acc, prod)
            outs.append(acc)
        return outs

    def decryptOutputs(self, evaluator, encoder, secret_key_dev, outputs):
        """-> [batchSize][outputChannels][H-kh+1][W-kw+1] doubles"""
        interval = self.blockHeight * self.blockWidth
        cs = (self.slotCount * 2) // interval
        yh, yw = self.blockHeight - self.kernelHeight + 1, self.blockWidth - self.kernelWidth + 1
        oyh, oyw = self.imageHeight - self.kernelHeight + 1, self.imageWidth - self.kernelWidth + 1
        sh, sw = self._grid() if self.blocked else (1, 1)
        ret = np.zeros((self.batchSize, self.outputChannels, oyh, oyw))
        for c in range(self.outputChannels):
            pts = evaluator.decrypt(outputs[c], secret_key_dev)
            for b in range(pts.shape[0]):
                buf = encoder.decodePolynomial(pts[b], outputs[c].scale)
                blk = buf[(cs - 1) * interval: cs * interval].reshape(self.blockHeight, self.blockWidth)[self.blockHeight - yh:, self.blockWidth - yw:]
                ob, si, sj = b // (sh * sw), (b % (sh * sw)) // sw, b % sw
                i1, j1 = min(si * yh + yh, oyh), min(sj * yw + yw, oyw)
                if si * yh < oyh and sj * yw < oyw:
                    ret[ob, c, si * yh: i1, sj * yw: j1] = blk[: i1 - si * yh, : j1 - sj * yw]
        return ret

    def _required(self, cid):  # LinearHelperCKKS.cuh:684-690: the last channel slot of every output polynomial
        interval = self.blockHeight * self.blockWidth
        st = ((self.slotCount * 2) // interval - 1) * interval
        return list(range(st, st + interval))

    def serializeOutputs(self, evaluator, outputs, stream):  # LinearHelperCKKS.cuh:684-696
        _serialize_terms(evaluator, outputs, self._required, stream)

    def deserializeOutputs(self, evaluator, context, stream):  # LinearHelperCKKS.cuh:698-713
        return _deserialize_terms(evaluator, context, self.getTotalBatchSize(), self.outputChannels, self._required, stream)
