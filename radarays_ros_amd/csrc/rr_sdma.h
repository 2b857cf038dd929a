// rr_sdma.h -- device -> page-locked host copies over the SDMA engines through ROCr, independent of the HIP runtime's choice of
// copy engine (rr_sdma.cpp says why and how).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

namespace rr {

struct SdmaCopier;
// any_device_ptr: a live device allocation of `hip_device` (identifies the HSA runtime instance and the GPU agent).  nullptr + why on failure
SdmaCopier* sdma_create(int hip_device, const void* any_device_ptr, std::string& why);
void sdma_destroy(SdmaCopier* s);                       // the copy in flight completes, queued ones are dropped
// copy `bytes` from d_src to h_dst once `after` (a recorded HIP event) has completed; jobs run in order.  Returns the job's id
uint64_t sdma_submit(SdmaCopier* s, hipEvent_t after, const void* d_src, void* h_dst, size_t bytes);
bool sdma_done(SdmaCopier* s, uint64_t job);
void sdma_wait(SdmaCopier* s, uint64_t job);
void sdma_wait_all(SdmaCopier* s);
// a job could not go over SDMA (it was delivered by a blocking hipMemcpy instead): the caller stops using the path
bool sdma_failed(SdmaCopier* s, std::string* why);

}  // namespace rr
