// custom_mm — pybind11 module with the reference's 16 entry points
// (smoorjani/matrix-multiplication src/custom_mm.cpp:393-416, same names, same
// positional signatures, same "caller allocates C, callee fills it and returns
// the same tensor" convention), backed by the MI355X C-ABI library
// libmi_spmm.so (include/mi_spmm.h).
//
// This layer only validates tensors, extracts pointers / leading dimensions,
// looks up torch's current HIP stream for the tensors' device and calls the
// C-ABI.  There is no CPU fallback: every compute entry point requires device
// tensors and raises otherwise.
//
// Differences from the reference that are deliberate:
//  * work is enqueued on torch's current stream of the tensors' device, not on
//    the legacy default stream (reference naive_sparse_mm.cu:117,133);
//  * dtype, device, shape and layout are checked (the reference reads
//    data_ptr() of whatever it is given, README.md:44); strided / transposed
//    views are handled through leading dimensions or copied, never misread.
//    The CONTENTS of CSR arrays (columns in range, offsets monotone) are trusted on
//    the per-product entry points, as in the reference; `validate_csr` checks them on
//    request and the inspector entry points check them once at inspect time;
//  * the inspect-style registries hold tensor references (the reference keeps
//    raw pointers and later cudaFree()s torch memory, custom_mm.cpp:249-251,
//    :272-277), are mutex-protected, and an unknown layer name raises instead
//    of aliasing handle 0 (custom_mm.cpp:260-263, :338-341).
#include <torch/extension.h>

#include <c10/hip/HIPGuard.h>
#include <c10/hip/HIPStream.h>

#include <cstdlib>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <tuple>
#include <unordered_map>

#include "mi_spmm.h"

namespace {
#include "custom_mm_helpers.inc"
#include "custom_mm_reference.inc"
#include "custom_mm_extras.inc"
#include "custom_mm_inspect.inc"

// ---- handle init / destroy (reference custom_mm.cpp:361-391) ----------------
// There are no vendor handles on this path; init checks that the C-ABI library
// this module was linked against has the expected ABI (printing to stderr on
// mismatch, like the reference's init failures) and all four are idempotent.
void init_backend(const char* name) {
  if (mi_spmm_abi_version() != MI_SPMM_ABI_VERSION)
    std::cerr << name << " initialization error: libmi_spmm ABI " << mi_spmm_abi_version()
              << " != " << MI_SPMM_ABI_VERSION << std::endl;
}
void init_cublas_handle() { init_backend("cuBLAS-path"); }
void destroy_cublas_handle() {}
void init_cusparse_handle() { init_backend("cuSPARSE-path"); }
void destroy_cusparse_handle() {}

// reference baseline_mm.cu:24-35 prints every thread id of a 64×64 launch;
// this launches the same grid, checks the ids on the host and prints one line.
void dummy_kernel_launch() {
  const auto dev = torch::Device(torch::kCUDA, c10::hip::current_device());
  torch::Tensor out = torch::full({4096}, -1, torch::dtype(torch::kInt32).device(dev));
  check_status(mi_dummy_kernel(out.data_ptr<int32_t>(), stream_of(out)), "dummy_kernel");
  torch::Tensor host = out.cpu();  // synchronises, like the reference's cudaDeviceSynchronize
  const bool ok = host.equal(torch::arange(4096, torch::kInt32));
  std::cout << "dummy_kernel: 64 blocks x 64 threads ran, ids " << (ok ? "0..4095 ok" : "WRONG")
            << std::endl;
  TORCH_CHECK(ok, "dummy_kernel wrote wrong thread ids");
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.def("init_cublas", &init_cublas_handle, "Create cuBLAS handle.");
  m.def("destroy_cublas", &destroy_cublas_handle, "Destroy cuBLAS handle.");

  m.def("init_cusparse", &init_cusparse_handle, "Create cuSPARSE handle.");
  m.def("destroy_cusparse", &destroy_cusparse_handle, "Destroy cuSPARSE handle.");

  m.def("cublas_mmul", &cublas_mmul, "cuBLAS Torch Matrix Multiplication");
  m.def("cublas_bmm", &cublas_bmm, "cuBLAS Batched Torch Matrix Multiplication");
  m.def("cusparse_mmul", &cusparse_mmul, "cuSPARSE Torch Matrix Multiplication");

  m.def("dummy_kernel", &dummy_kernel_launch, "Launch dummy kernel.");
  m.def("naive_spmm", &naive_spmm, "A naive implementation of Sparse Matrix Multiplication");

  m.def("tiledspmm_inspect_csr", &tiledspmm_inspect_csr, "Inspect function for TiledSpMM with CSR input");
  m.def("tiledspmm_inspect_coo", &tiledspmm_inspect_coo, "Inspect function for TiledSpMM with COO input");
  m.def("tiledspmm_mm", &tiledspmm_mm, "MM function for TiledSpMM");
  m.def("tiledspmm_clean", &tiledspmm_clean, "Cleanup function for TiledSpMM");

  m.def("cusparse_inspect", &cusparse_inspect, "Inspect function for CuSPARSE with CSR input");
  m.def("cusparse_mmul_opt", &cusparse_mmul_opt, "MM function for CuSPARSE");
  m.def("cusparse_clean", &cusparse_clean, "Cleanup function for CuSPARSE");
  m.def("cusparse_mmul_opt_t", &cusparse_mmul_opt_t, "column-major product with the cached transpose: C = A^T B");
  m.def("tiledspmm_mm_t", &tiledspmm_mm_t, "column-major product with the cached transpose: C = A^T B");
  m.def("inspect_info", &inspect_info, "what an inspector handle holds (layer, tiled)");

  // additions (not in the reference): one-launch batching and the sparse backward
  m.def("dense_to_csr", &dense_to_csr, "Device dense -> batched CSR (values, columns, offsets)");
  m.def("dense_row_offsets", &dense_row_offsets, "Device dense -> row offsets [batch, rows+1] only (no host sync)");
  m.def("dense_to_csr_fill", &dense_to_csr_fill, "(values, columns) for the offsets of dense_row_offsets");
  m.def("naive_spmm_batched", &naive_spmm_batched, "Batched CSR x dense in one launch");
  m.def("csr_transpose", &csr_transpose, "Device CSR transpose (values, columns, offsets)");
  m.def("csr_transpose_batched", &csr_transpose_batched, "Batched device CSR transpose (values, columns, offsets [batch, cols+1])");
  m.def("csr_transpose_in_lds", [](int64_t nnz, int64_t batch, int64_t rows, int64_t cols) {
          return rows <= INT32_MAX && cols <= INT32_MAX && batch <= INT32_MAX &&
                 mi_csr_transpose_batched_in_lds(nnz, (int32_t)batch, (int32_t)rows, (int32_t)cols) == 1;
        }, "does csr_transpose_batched run the one-workgroup-per-item LDS plan on this batch (cheap: transpose per backward)");
  m.def("sddmm", &sddmm, "Sampled dense-dense product on a CSR pattern");
  m.def("cublas_bmm_pair", &cublas_bmm_pair,
        "dA = dC.B and dB = dC^T.A in one launch that reads dC once; (dC, B, A, dA, dB) -> launched?");
  m.def("sddmm_batched", &sddmm_batched,
        "SDDMM on a batched CSR pattern into out[nnz]; False (nothing launched) if the LDS-resident form does not take it");
  m.def("gather_perm", &gather_perm, "values[perm] (int32 perm) as a new tensor");
  m.def("naive_spmm_batched_perm", &naive_spmm_batched_perm,
        "naive_spmm_batched with entry p's value = A_values[perm[p]]; False (nothing launched) if the plan takes no permutation");
  m.def("naive_spmm_batched_at", &naive_spmm_batched_at,
        "C[i] = A[i]^T X[i] for a batched CSR A without transposing it; False (nothing launched) if the shape is not covered");
  m.def("batched_csr_narrow", &batched_csr_narrow,
        "(int32 offsets [batch, rows+1] with the items' bases, int32 columns) of int64 crow [batch, rows+1] / col [batch, per_item]");
  m.def("naive_spmm_dense", &naive_spmm_dense,
        "A·B with A dense, zeros skipped in the kernel; False if the shape is not covered");
  m.def("naive_spmm_dense_bias", &naive_spmm_dense_bias, "as naive_spmm_dense, + bias in the epilogue");
  m.def("nonfinite_flag", &nonfinite_flag, "int32[1] device tensor: 1 if the tensor holds an inf / nan (nothing read back)");
  m.def("naive_spmm_dense_gated", &naive_spmm_dense_gated,
        "naive_spmm_dense as a launch that runs only when flag[0] != 0 on the device; (A, B, C, flag, dry_run) -> covered?");
  m.def("cublas_mmul_bias", &cublas_mmul_bias, "op(A) op(B) + bias, fused epilogue");
  m.def("column_sums", &column_sums, "sum over rows of a 2-d tensor (bias gradient)");
  m.def("naive_spmm_bias", &naive_spmm_bias, "CSR x dense + bias, fused epilogue");
  m.def("naive_spmm_ex", &naive_spmm_ex, "naive_spmm with the long-row rule pinned (-1 auto, 0 none, 1 split)");
  m.def("naive_spmm_bias_ex", &naive_spmm_bias_ex, "naive_spmm_bias with the long-row rule pinned (-1 auto, 0 none, 1 split)");
  m.def("spmm_plan", &spmm_plan, "(variant, kernel name, launches, splits_long_rows) of the AUTO plan");
  m.def("validate_csr", &validate_csr, "opt-in check of CSR contents (offsets monotone, columns in range); raises");
  py::class_<PySchedule, std::shared_ptr<PySchedule>>(m, "SpmmSchedule")
      .def("info", &PySchedule::info, "rows, heavy_rows, heavy_length, classes, longest_at_least, side_stream, active, locality_order, window spans, nnz, width")
      .def("set_heavy", &PySchedule::set_heavy, py::arg("heavy_length"), py::arg("side_stream") = true,
           "another heavy length / the heavy launch in line (tests, A/B)")
      .def_readonly("order", &PySchedule::order, "int32 [rows] on the device: slot -> row, rows by descending length class");
  m.def("spmm_schedule", &spmm_schedule, py::arg("A_offsets"), py::arg("nnzA"), py::arg("A_rows"), py::arg("N"),
        py::arg("A_columns") = py::none(), py::arg("A_cols") = 0,
        "Inspector: the row schedule of a CSR matrix (offsets, nnz, rows, dense width; with the columns: also tries the locality order)");
  m.def("naive_spmm_scheduled", &naive_spmm_scheduled, py::arg("schedule"), py::arg("A_values"), py::arg("A_columns"),
        py::arg("A_offsets"), py::arg("nnzA"), py::arg("A_rows"), py::arg("A_cols"), py::arg("B"), py::arg("C"),
        py::arg("bias") = py::none(), py::arg("long_rows") = -1, py::arg("variant") = 0,
        "naive_spmm on a row schedule: the same bits, rows handed to waves longest first");
  m.def("auto_schedule_stats", &auto_schedule_stats, "the automatic row schedules of the plain entry points: enabled, entries, pending, active, inactive, built");
  m.def("auto_schedule_clear", &auto_schedule_clear, "forget every automatic row schedule");
  m.def("ipc_export", &ipc_export, "(handle bytes, offset, allocation bytes) of a device buffer, for a peer process to map");
  m.def("ipc_open", &ipc_open, "a peer's exported buffer as a tensor (valid until ipc_close of the handle)");
  m.def("ipc_close", &ipc_close, "drop one open of a peer handle");
  m.def("ipc_open_count", []() { return mi_ipc_open_count(); }, "peer mappings not yet closed in this process");
  m.def("long_row_threshold", &long_row_threshold, "rows with more non-zeros are 'long' (split rule)");
  // Handles and automatic schedules own HIP streams and events: they are released while the interpreter — and with it the HIP
  // runtime — is still up (left to the destructors of the statics they segfaulted at process exit after the runtime had gone:
  // a program that never called cusparse_clean / auto_schedule_clear ended with exit code 139 AFTER its last line of output).
  py::module_::import("atexit").attr("register")(py::cpp_function([]() {
    auto_schedule_clear();
    std::lock_guard<std::mutex> lock(g_registry_mutex);
    g_cusparse_layers.clear();
    g_tiled_layers.clear();
  }));
}
