// TEST INFRASTRUCTURE (CPU baseline of the reference's own benchmark): stands where src/troy_cuda.cuh stands when the reference's test/timetest.cu is
// compiled against the reference's CPU half (src/troy_cpu.h, oracle/_ref/libtroyref.so) instead of its CUDA half -- the *Cuda names become the CPU
// classes of the same interface (macros: the CPU headers forward-declare the *Cuda classes as friends, so aliases of those names would clash).
// oracle/Makefile feeds test/timetest.cu through stdin from oracle/ref_shim/test/, so its #include "../src/troy_cuda.cuh" lands here; nothing of the
// reference is copied.  The result, oracle/_ref/ref_timetest, is the CPU column of profiles/r05_timetest.txt.
#pragma once
#include <complex>
#include <iomanip>
#include <iostream>
#include <map>
#include <string>
#include <vector>
#include "troy_cpu.h"

namespace troy {
    struct KernelProviderCpu { static void initialize(int = 0) {} };
}
#define KernelProvider KernelProviderCpu
#define EncryptionParametersCuda EncryptionParameters
#define SEALContextCuda SEALContext
#define PlaintextCuda Plaintext
#define CiphertextCuda Ciphertext
#define EncryptorCuda Encryptor
#define DecryptorCuda Decryptor
#define EvaluatorCuda Evaluator
#define KeyGeneratorCuda KeyGenerator
#define PublicKeyCuda PublicKey
#define SecretKeyCuda SecretKey
#define KSwitchKeysCuda KSwitchKeys
#define RelinKeysCuda RelinKeys
#define GaloisKeysCuda GaloisKeys
#define CKKSEncoderCuda CKKSEncoder
#define BatchEncoderCuda BatchEncoder
namespace troyn {
    using troy::ParmsID; using troy::SchemeType; using troy::SecurityLevel; using troy::Modulus; using troy::CoeffModulus; using troy::PlainModulus;
    using EncryptionParameters = troy::EncryptionParameters; using SEALContext = troy::SEALContext; using Plaintext = troy::Plaintext;
    using Ciphertext = troy::Ciphertext; using Encryptor = troy::Encryptor; using Decryptor = troy::Decryptor; using Evaluator = troy::Evaluator;
    using KeyGenerator = troy::KeyGenerator; using PublicKey = troy::PublicKey; using SecretKey = troy::SecretKey; using KSwitchKeys = troy::KSwitchKeys;
    using RelinKeys = troy::RelinKeys; using GaloisKeys = troy::GaloisKeys; using CKKSEncoder = troy::CKKSEncoder; using BatchEncoder = troy::BatchEncoder;
    using KernelProvider = troy::KernelProviderCpu;
}
