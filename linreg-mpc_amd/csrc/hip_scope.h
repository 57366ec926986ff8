// hip_scope.h -- temporary device buffers of a host function: released on every return path
#pragma once
#include <hip/hip_runtime.h>
#include <vector>
struct DevFree {
    std::vector<void *> owned;                       // pointer VALUES (no references to locals)
    std::vector<size_t> wipe;                        // bytes to clear before the free (buffers that held label pairs / OT rows)
    void add(void *p) { if (p) { owned.push_back(p); wipe.push_back(0); } }
    void add_secret(void *p, size_t bytes) { if (p) { owned.push_back(p); wipe.push_back(bytes); } }
    void release(void *p) {                          // ownership passes to someone else
        for (size_t i = 0; i < owned.size(); i++) if (owned[i] == p) { owned[i] = 0; }
    }
    ~DevFree() {
        for (size_t i = 0; i < owned.size(); i++) {
            if (!owned[i]) continue;
            if (wipe[i]) (void)hipMemset(owned[i], 0, wipe[i]);
            (void)hipFree(owned[i]);
        }
    }
};
