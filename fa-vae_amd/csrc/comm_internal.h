// Internal view of favae_comm (comm.hip) for the executor's data-parallel epoch (vae_model.hip).
#pragma once
#include <hip/hip_runtime.h>
struct favae_comm;
hipStream_t favae_comm_stream(favae_comm* c);
hipEvent_t favae_comm_event(favae_comm* c);
int favae_comm_size(favae_comm* c);
