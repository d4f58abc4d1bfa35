// exchange.hpp — multi-GPU exchange steps (exchange.hip): all-gatherv and hash repartition of binding tables.
#pragma once
#include "common.hpp"
#include "kernels.hpp"

namespace rdfgpu {

struct Comm;
void comm_unique_id(unsigned char id[128]);
Comm* comm_create_rccl(const unsigned char id[128], u32 rank, u32 world, int device);
Comm* comm_create_host(u32 rank, u32 world, int device, rdfgpu_host_alltoallv_fn fn, void* ctx);
void comm_destroy(Comm* c);
u32 comm_rank(const Comm* c);
u32 comm_world(const Comm* c);
// both return the number of rows received; out_cols[c] are communicator-owned device columns, valid until the next exchange
u64 exchange_allgatherv(Comm* c, const u32* const* cols, u32 n_cols, u64 n_rows, const u32** out_cols);
u64 exchange_repartition(Comm* c, const u32* const* cols, u32 n_cols, u64 n_rows, u32 key_col, const u32** out_cols);

}  // namespace rdfgpu
