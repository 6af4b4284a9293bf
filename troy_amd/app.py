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


def _ceil_div(a, b):
    return (a + b - 1) // b


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
            cts = np.stack([encryptor.encrypt(encoder.encodePolynomial(X[b, lj:uj], limbs, scale)) for b in range(self.batchSize)])
            out.append(api.Ciphertext.from_numpy(ctx, cts, True, scale, 1))
        return out

    def matmul(self, evaluator, a):
        """Cipher2d x encoded weights -> list over output blocks of one batched ciphertext (same order of additions as the
        reference: block row i = 0 initialises, the others are added in order)"""
        if len(a) != len(self.encodedWeights):
            raise capi.InvalidArgument(capi.INVALID_ARGUMENT, "Input size incorrect.")
        outs = [None] * len(self.encodedWeights[0])
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
