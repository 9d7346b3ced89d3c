// The compiled module the reference imports as ``fcos_core._C`` (reference fcos_core/csrc/vision.cpp:8-17, built by its
// setup.py:41-50 from csrc/{cpu,cuda}/*): the same function names, at::Tensor signatures, argument meaning and error
// behaviour, for the four functions the SCAN hot path calls -- backed by libscan_hip.so through the C ABI of
// include/scan_hip.h (plain pointers, sizes and a hipStream_t; nothing of torch crosses that boundary).
//
//   nms(dets [n,4], scores [n], threshold[, cuda_rule]) -> int64 [k]                       csrc/nms.h:10-30
//     GPU tensors: libscan_hip.so; CPU tensors (float / double): the host loop below           csrc/cpu/nms_cpu.cpp:5-75
//     tie rule: IoU >= threshold suppresses (nms_cpu.cpp:60, what tests/test_nms.py pins) unless cuda_rule = True or
//     SCAN_NMS_RULE=gt is set in the environment: then IoU > threshold (csrc/cuda/nms.cu:60)
//   ml_nms(dets [n,4], scores [n], labels [n] float, threshold) -> int64 [k]              csrc/ml_nms.h:10-29
//   sigmoid_focalloss_forward(logits [M,C], targets [M] int32, C, gamma, alpha) -> [M,C]  csrc/SigmoidFocalLoss.h:10-24
//   sigmoid_focalloss_backward(logits, targets, d_losses, C, gamma, alpha) -> [M,C]       csrc/SigmoidFocalLoss.h:26-41
//   roi_align_* / roi_pool_*: two-stage heads no SCAN config uses: raise.
//
// Built by __graft_entry__.build() (plain g++: no device code in this file) into
// scan_amd/ext/fcos_core/_C<EXT_SUFFIX>; with scan_amd/ext on sys.path, ``from fcos_core import _C`` is this module and
// the reference's layers/nms.py:4-7 and layers/sigmoid_focal_loss.py:9-36 run on it unchanged.
#include <ATen/ATen.h>
// PyTorch-ROCm tensors carry the device type "cuda": guard and stream come from the masquerading wrappers (the plain
// c10::hip::HIPGuard insists on DeviceType::HIP and throws on them)
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <torch/extension.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <vector>

#include "../../include/scan_hip.h"

namespace {

void* current_stream(const at::Tensor& t) { return c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.device().index()).stream(); }

void check(int rc, const char* what) { TORCH_CHECK(rc == 0, what, " failed (", rc, "): ", scan_last_error()); }

// Greedy NMS on host memory, the reference's CPU semantics (csrc/cpu/nms_cpu.cpp:5-65): candidates by descending score
// (equal scores: lower index first), a kept box suppresses every later one whose IoU (areas with the +1 pixel rule, in
// the tensors' own precision) reaches the threshold; kept ORIGINAL indices ascending.
template <typename T>
at::Tensor nms_host(const at::Tensor& dets_in, const at::Tensor& scores_in, float threshold, bool rule_ge) {
  const at::Tensor dets = dets_in.contiguous(), scores = scores_in.contiguous();
  const int64_t n = dets.size(0);
  const T* box = dets.data_ptr<T>();
  const T* sc = scores.data_ptr<T>();
  std::vector<int64_t> by_score(n);
  std::iota(by_score.begin(), by_score.end(), int64_t{0});
  std::stable_sort(by_score.begin(), by_score.end(), [sc](int64_t a, int64_t b) { return sc[a] > sc[b]; });
  std::vector<T> area(n);
  for (int64_t i = 0; i < n; ++i) area[i] = (box[4 * i + 2] - box[4 * i] + 1) * (box[4 * i + 3] - box[4 * i + 1] + 1);
  std::vector<char> dead(n, 0);
  for (int64_t r = 0; r < n; ++r) {
    const int64_t i = by_score[r];
    if (dead[i]) continue;
    const T* a = box + 4 * i;
    for (int64_t q = r + 1; q < n; ++q) {
      const int64_t j = by_score[q];
      if (dead[j]) continue;
      const T* b = box + 4 * j;
      const T w = std::max(T(0), std::min(a[2], b[2]) - std::max(a[0], b[0]) + 1);
      const T h = std::max(T(0), std::min(a[3], b[3]) - std::max(a[1], b[1]) + 1);
      const T inter = w * h;
      const T iou = inter / (area[i] + area[j] - inter);
      if (rule_ge ? iou >= threshold : iou > threshold) dead[j] = 1;
    }
  }
  std::vector<int64_t> kept;
  for (int64_t i = 0; i < n; ++i)
    if (!dead[i]) kept.push_back(i);
  at::Tensor out = at::empty({(int64_t)kept.size()}, at::TensorOptions().dtype(at::kLong).device(at::kCPU));
  if (!kept.empty()) std::memcpy(out.data_ptr<int64_t>(), kept.data(), kept.size() * sizeof(int64_t));
  return out;
}

at::Tensor nms_impl(const at::Tensor& dets_in, const at::Tensor& scores_in, const at::Tensor* labels_in, float threshold,
                    int rule_ge, const char* who) {
  if (dets_in.numel() == 0)  // csrc/nms.h:17-18: an empty CPU int64 tensor whatever the device
    return at::empty({0}, dets_in.options().dtype(at::kLong).device(at::kCPU));
  TORCH_CHECK(dets_in.dim() == 2 && dets_in.size(1) == 4, who, ": dets must be [n, 4]");
  const int64_t n = dets_in.size(0);
  TORCH_CHECK(scores_in.numel() == n, who, ": scores must have one entry per box");
  if (!dets_in.is_cuda()) {
    TORCH_CHECK(labels_in == nullptr, who, ": not implemented on the CPU");  // csrc/ml_nms.h:26
    TORCH_CHECK(!scores_in.is_cuda(), "scores must be a CPU tensor");                                   // nms_cpu.cpp:10
    TORCH_CHECK(dets_in.scalar_type() == scores_in.scalar_type(), "dets should have the same type as scores");  // :11
    if (dets_in.scalar_type() == at::kDouble) return nms_host<double>(dets_in, scores_in, threshold, rule_ge != 0);
    TORCH_CHECK(dets_in.scalar_type() == at::kFloat, who, ": float or double boxes (AT_DISPATCH_FLOATING_TYPES, nms_cpu.cpp:71)");
    return nms_host<float>(dets_in, scores_in, threshold, rule_ge != 0);
  }
  TORCH_CHECK(n <= SCAN_NMS_MAX, who, ": n=", n, " exceeds SCAN_NMS_MAX=", SCAN_NMS_MAX);
  c10::hip::HIPGuardMasqueradingAsCUDA guard(dets_in.device());
  const at::Tensor dets = dets_in.contiguous().to(at::kFloat), scores = scores_in.contiguous().to(at::kFloat);
  at::Tensor labels;
  if (labels_in != nullptr) {
    TORCH_CHECK(labels_in->numel() == n, who, ": labels must have one entry per box");
    labels = labels_in->contiguous().to(at::kFloat);
  }
  at::Tensor ws = at::empty({(scan_nms_ws_bytes(n) + 15) / 16 * 2}, dets.options().dtype(at::kDouble));
  at::Tensor keep = at::empty({n}, dets.options().dtype(at::kLong));
  at::Tensor cnt = at::empty({1}, dets.options().dtype(at::kInt));
  check(scan_nms(dets.data_ptr<float>(), scores.data_ptr<float>(), labels.defined() ? labels.data_ptr<float>() : nullptr, n,
                 threshold, rule_ge, keep.data_ptr<int64_t>(), cnt.data_ptr<int32_t>(), ws.data_ptr(), current_stream(dets)),
        who);
  return keep.slice(0, 0, cnt.item<int32_t>());  // kept ORIGINAL indices, ascending (csrc/cpu/nms_cpu.cpp:64)
}

bool env_cuda_rule() {
  const char* e = std::getenv("SCAN_NMS_RULE");
  return e != nullptr && (std::strcmp(e, "gt") == 0 || std::strcmp(e, "cuda") == 0);
}

// Default: IoU >= threshold suppresses -- the reference's CPU rule (csrc/cpu/nms_cpu.cpp:60), which its own tests/test_nms.py
// pins.  cuda_rule = true (or SCAN_NMS_RULE=gt): IoU > threshold, what the reference's GPU build computes (cuda/nms.cu:60).
at::Tensor nms(const at::Tensor& dets, const at::Tensor& scores, const float threshold, const bool cuda_rule) {
  return nms_impl(dets, scores, nullptr, threshold, (cuda_rule || env_cuda_rule()) ? 0 : 1, "nms");
}

// label-aware, IoU > threshold suppresses: the reference's CUDA rule (csrc/cuda/ml_nms.cu:13-24,62); no CPU version exists
at::Tensor ml_nms(const at::Tensor& dets, const at::Tensor& scores, const at::Tensor& labels, const float threshold) {
  return nms_impl(dets, scores, &labels, threshold, 0, "ml_nms");
}

at::Tensor sigmoid_focalloss_forward(const at::Tensor& logits_in, const at::Tensor& targets_in, const int num_classes,
                                     const float gamma, const float alpha) {
  TORCH_CHECK(logits_in.is_cuda(), "Not implemented on the CPU");              // csrc/SigmoidFocalLoss.h:23
  TORCH_CHECK(targets_in.is_cuda(), "targets must be a CUDA tensor");          // SigmoidFocalLoss_cuda.cu:110
  TORCH_CHECK(logits_in.dim() == 2, "logits should be NxClass");               // :112
  TORCH_CHECK(logits_in.size(1) == num_classes, "logits.size(1) should be num_classes");
  TORCH_CHECK(targets_in.scalar_type() == at::kInt, "targets must be int32 (the reference passes targets.int())");
  c10::hip::HIPGuardMasqueradingAsCUDA guard(logits_in.device());
  const at::Tensor logits = logits_in.contiguous().to(at::kFloat), targets = targets_in.contiguous();
  at::Tensor losses = at::empty_like(logits);
  check(scan_sigmoid_focal_loss_forward(logits.data_ptr<float>(), targets.data_ptr<int32_t>(), logits.size(0), num_classes,
                                        gamma, alpha, losses.data_ptr<float>(), nullptr, current_stream(logits)),
        "sigmoid_focalloss_forward");
  return losses;
}

at::Tensor sigmoid_focalloss_backward(const at::Tensor& logits_in, const at::Tensor& targets_in, const at::Tensor& d_losses_in,
                                      const int num_classes, const float gamma, const float alpha) {
  TORCH_CHECK(logits_in.is_cuda(), "Not implemented on the CPU");              // csrc/SigmoidFocalLoss.h:39
  TORCH_CHECK(targets_in.is_cuda() && d_losses_in.is_cuda(), "targets and d_losses must be CUDA tensors");
  TORCH_CHECK(logits_in.dim() == 2 && logits_in.size(1) == num_classes, "logits.size(1) should be num_classes");  // :158
  TORCH_CHECK(targets_in.scalar_type() == at::kInt, "targets must be int32 (the reference passes targets.int())");
  TORCH_CHECK(d_losses_in.sizes() == logits_in.sizes(), "d_losses must have the shape of logits");
  c10::hip::HIPGuardMasqueradingAsCUDA guard(logits_in.device());
  const at::Tensor logits = logits_in.contiguous().to(at::kFloat), targets = targets_in.contiguous(),
                   d_losses = d_losses_in.contiguous().to(at::kFloat);
  at::Tensor d_logits = at::empty_like(logits);
  check(scan_sigmoid_focal_loss_backward(logits.data_ptr<float>(), targets.data_ptr<int32_t>(), d_losses.data_ptr<float>(),
                                         1.0f, logits.size(0), num_classes, gamma, alpha, d_logits.data_ptr<float>(),
                                         current_stream(logits)),
        "sigmoid_focalloss_backward");
  return d_logits;
}

at::Tensor two_stage(py::args, py::kwargs) {
  TORCH_CHECK(false, "roi_align / roi_pool are outside the SCAN hot path (RPN_ONLY configs); not built");
  return at::Tensor();
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.doc() = "fcos_core._C on libscan_hip.so (MI355X / gfx950)";
  m.def("nms", &nms, "non-maximum suppression", py::arg("dets"), py::arg("scores"), py::arg("threshold"),
        py::arg("cuda_rule") = false);
  m.def("ml_nms", &ml_nms, "multi-label non-maximum suppression");
  m.def("sigmoid_focalloss_forward", &sigmoid_focalloss_forward, "SigmoidFocalLoss_forward");
  m.def("sigmoid_focalloss_backward", &sigmoid_focalloss_backward, "SigmoidFocalLoss_backward");
  m.def("roi_align_forward", &two_stage, "ROIAlign_forward");
  m.def("roi_align_backward", &two_stage, "ROIAlign_backward");
  m.def("roi_pool_forward", &two_stage, "ROIPool_forward");
  m.def("roi_pool_backward", &two_stage, "ROIPool_backward");
  m.def("scan_abi_version", []() { return scan_abi_version(); }, "ABI version of the libscan_hip.so this module is linked to");
}
