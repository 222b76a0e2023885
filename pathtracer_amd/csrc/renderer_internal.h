// What the communication layer (pt_comm.cpp) needs from a renderer without seeing its layout.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <string>

struct pt_renderer;

namespace pt_internal {

struct SumBuffer {
    float *data;          // width * height * 4 floats on `device`
    size_t count;         // number of floats
    int device;
    hipStream_t stream;   // the renderer's main stream
};

// Waits for everything the renderer has enqueued (both batch streams), then describes its sum buffer.
// Returns false (with pt_last_error set) when the renderer has no image buffers.
bool sum_buffer(pt_renderer *r, SumBuffer *out);
// Work was enqueued on the main stream that the second batch stream's next batch must wait for.
void main_stream_work_enqueued(pt_renderer *r);
// Records `msg` as the calling thread's last error and returns `code`.
int set_error(int code, const std::string &msg);

}  // namespace pt_internal
