// Internal view of favae_comm (comm.hip) for the executor's data-parallel epoch (vae_model.hip).
#pragma once
#include <hip/hip_runtime.h>
struct favae_comm;
int favae_comm_size(favae_comm* c);
