// Host-side launch interface of the loss kernels (prototype losses).  See include/vqseg.h for the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vqseg {

struct ProtoArgs {
    const void* x;               // decoder features, rows [M][C] (f32 or bf16)
    int bf16;
    const float* proto;          // L2-normalised prototypes [K][C]
    const long long* labels;     // [M] class ids
    const unsigned char* keep;   // v1: [M] entropy filter (nullable = all kept)
    const float* conf;           // v2: [M] confidence weights (nullable = 1)
    long M;
    int C, K, variant;           // variant 1: ReliablePrototypeLoss, 2: ReliablePrototypeLossv2
    float scale, cos_m, sin_m, th, mm;
    int easy_margin, use_margin; // use_margin: v1 applies phi only when margin != 0
};

constexpr int PROTO_ROWS_PER_BLOCK = 256;
long proto_blocks(long M);
hipError_t launch_proto_forward(const ProtoArgs& a, double* partial, double* loss, hipStream_t st);
hipError_t launch_proto_backward(const ProtoArgs& a, const float* g_loss, void* gx, float* gproto_partial, float* gproto,
                                 hipStream_t st);

}  // namespace vqseg
