// TEST INFRASTRUCTURE (see ../hip/hip_runtime.h): the hipCUB entry points the product calls.
#ifndef WXA_TESTS_HIPCPU_HIPCUB_HPP_
#define WXA_TESTS_HIPCPU_HIPCUB_HPP_
#include <hip/hip_runtime.h>

namespace hipcub {
struct DeviceScan {
    template <class In, class Out>
    static hipError_t ExclusiveSum(void* tmp, size_t& tmp_bytes, In in, Out out, int n, hipStream_t = nullptr) {
        if (!tmp) { tmp_bytes = 1; return hipSuccess; }
        std::remove_cv_t<std::remove_reference_t<decltype(out[0])>> run = 0;
        for (int i = 0; i < n; ++i) { const auto v = in[i]; out[i] = run; run += v; }
        return hipSuccess;
    }
};
}  // namespace hipcub
#endif
