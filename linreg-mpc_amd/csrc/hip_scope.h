// hip_scope.h -- temporary device buffers of a host function: released on every return path
#pragma once
#include <hip/hip_runtime.h>
#include <vector>
struct DevFree {
    std::vector<void *> owned;                       // pointer VALUES (no references to locals)
    void add(void *p) { if (p) owned.push_back(p); }
    void release(void *p) {                          // ownership passes to someone else
        for (size_t i = 0; i < owned.size(); i++) if (owned[i] == p) { owned[i] = 0; }
    }
    ~DevFree() { for (size_t i = 0; i < owned.size(); i++) if (owned[i]) (void)hipFree(owned[i]); }
};
