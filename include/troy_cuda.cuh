// troy_cuda.cuh -- the header the reference's GPU callers include (/root/reference/src/troy_cuda.cuh:1-43), over the MI355X library.
// The reference defines its device classes in namespace troy (troy::EvaluatorCuda, troy::CiphertextCuda, ...) and aliases them into
// namespace troyn; its own tests and apps spell BOTH forms (test/evaluator_cuda.cu:10-34 the first, test/ckks_cuda.cu:9 and
// app/LinearHelper.cuh the second).  Here the classes live in troyn (include/troyn.hpp) and the troy::*Cuda names are the aliases.
#pragma once
#include <iostream>
#include "troyn.hpp"

namespace troy {
    using troyn::ParmsID;
    using troyn::parmsIDZero;
    using troyn::SchemeType;
    using troyn::SecurityLevel;
    using troyn::Modulus;
    using troyn::CoeffModulus;
    using troyn::PlainModulus;
    using troyn::KernelProvider;
    using EncryptionParametersCuda = troyn::EncryptionParameters;
    using SEALContextCuda = troyn::SEALContext;
    using PlaintextCuda = troyn::Plaintext;
    using CiphertextCuda = troyn::Ciphertext;
    using EncryptorCuda = troyn::Encryptor;
    using DecryptorCuda = troyn::Decryptor;
    using EvaluatorCuda = troyn::Evaluator;
    using KeyGeneratorCuda = troyn::KeyGenerator;
    using PublicKeyCuda = troyn::PublicKey;
    using SecretKeyCuda = troyn::SecretKey;
    using KSwitchKeysCuda = troyn::KSwitchKeys;
    using RelinKeysCuda = troyn::RelinKeys;
    using GaloisKeysCuda = troyn::GaloisKeys;
    using CKKSEncoderCuda = troyn::CKKSEncoder;
    using BatchEncoderCuda = troyn::BatchEncoder;
    using LWECiphertextCuda = troyn::LWECiphertext;
}
