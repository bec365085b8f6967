// amq_torch_ext.cpp -- host-side fast path of the drop-in modules: ONE call per forward, no ctypes marshalling.
//
// The reference's modules call pybind functions that take torch::Tensor arguments
//   (auto_gptq.vecquant{2,3,4}matmul_faster_old: hqq/backends/autogptq.py:171-243; faster_transformer.gemv_4bit / gemm_4bit:
//    hqq/backends/ft.py:129-145);
// this is the same level of binding over libamq_hip.so's C ABI (include/amq_hip.h): the tensor checks, the output
// allocation, the current-stream query and the library call happen in C++.  A HIPQuantLinear.forward through ctypes costs
// ~10 us on the host (tools/module_walk_bench.py); through here ~3 us.
// The library is NOT linked: init(path) dlopens it and resolves the entry points, so this file builds with g++ against the
// torch headers alone (no HIP compiler, __graft_entry__.build()).  Every function raises (never falls back) on a bad argument
// or a non-zero library status.
#include <torch/extension.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>     // (ROCm torch tensors carry DeviceType cuda: the masquerading guard is the one that accepts them)
#include <c10/hip/HIPStream.h>
#include <dlfcn.h>

#include <stdexcept>
#include <string>
#include <vector>

namespace {

struct Segment { const void* qweight; const void* meta; const void* bias; const void* residual; void* y; int N, bits, mode, y_stride; };
struct GemvOpts { int math, waves, depth, rpt, dot; };

using linear_fn = int (*)(int, int, const void*, const void*, const void*, const void*, void*, int, int, int, int, void*);
using grouped_fn = int (*)(const Segment*, int, const void*, const void*, const void*, float, int, int, int, int, int, const GemvOpts*, void*);
using rmsnorm_fn = int (*)(const void*, const void*, void*, int, int, float, void*);
using silu_mul_fn = int (*)(const void*, const void*, void*, size_t, void*);
using attn_cur_fn = int (*)(const void*, const void*, const void*, void*, void*, void*, const void*, int, int, int, int, int, void*);
using err_fn = const char* (*)();

void* g_lib = nullptr;
linear_fn g_linear = nullptr;
grouped_fn g_grouped = nullptr;
rmsnorm_fn g_rmsnorm = nullptr;
silu_mul_fn g_silu_mul = nullptr;
attn_cur_fn g_attn_cur = nullptr;
err_fn g_err = nullptr;

void need_lib() {
    if (!g_lib) throw std::runtime_error("amq torch extension: init(path to libamq_hip.so) has not been called");
}
void check_rc(int rc, const char* what) {
    if (rc != 0) throw std::runtime_error(std::string("libamq_hip ") + what + " failed (" + std::to_string(rc) + "): " + (g_err ? g_err() : ""));
}
void* stream_of(const at::Tensor& t) { return (void*)c10::hip::getCurrentHIPStream(t.device().index()).stream(); }

void check_x(const at::Tensor& x, int64_t K) {
    TORCH_CHECK(x.is_cuda() && x.scalar_type() == at::kHalf && x.is_contiguous(), "x: expected a contiguous fp16 tensor on the GPU");
    TORCH_CHECK(x.dim() >= 1 && x.size(-1) == K, "x: last dim ", x.size(-1), " != K=", K);
}
// returns the native meta's granularity: 128 (one (scale, zero) pair per tile row), or 64 / 32 (two / four pairs: amq_common.cuh)
int check_native(const at::Tensor& qw, const at::Tensor& mt, int64_t bits, int64_t N, int64_t K, const at::Tensor& x) {
    TORCH_CHECK(bits >= 2 && bits <= 4 && N % 16 == 0 && K % 128 == 0, "bits must be 2..4, N % 16 == 0, K % 128 == 0");
    TORCH_CHECK(qw.scalar_type() == at::kInt && qw.is_contiguous() && qw.numel() == N * K * bits / 32, "qweight: not a native payload of this shape");
    const int64_t base = N * (K / 128) * 2;
    TORCH_CHECK(mt.scalar_type() == at::kHalf && mt.is_contiguous() && (mt.numel() == base || mt.numel() == 2 * base || mt.numel() == 4 * base),
                "meta: not a native meta buffer of this shape");
    TORCH_CHECK(qw.device() == x.device() && mt.device() == x.device(), "x and the weights must be on the same device");
    return (int)(128 * base / mt.numel());
}

}  // namespace

void init(const std::string& path) {
    if (g_lib) return;
    void* h = dlopen(path.c_str(), RTLD_NOW | RTLD_GLOBAL);
    if (!h) throw std::runtime_error(std::string("amq torch extension: cannot load ") + path + ": " + dlerror());
    g_linear = (linear_fn)dlsym(h, "amq_linear_f16");
    g_grouped = (grouped_fn)dlsym(h, "amq_gemv_grouped_f16");
    g_rmsnorm = (rmsnorm_fn)dlsym(h, "amq_rmsnorm_f16");
    g_silu_mul = (silu_mul_fn)dlsym(h, "amq_silu_mul_f16");
    g_attn_cur = (attn_cur_fn)dlsym(h, "amq_attn_decode_cur_f16");
    g_err = (err_fn)dlsym(h, "amq_last_error");
    if (!g_linear || !g_grouped || !g_rmsnorm || !g_silu_mul || !g_attn_cur || !g_err) throw std::runtime_error("amq torch extension: libamq_hip.so lacks an entry point");
    g_lib = h;
}

// y[..., N] = x[..., K] . W^T (+ bias) for M = numel / K <= 8 rows (amq_linear_f16: the weight-streaming GEMV)
at::Tensor linear(const at::Tensor& x, const at::Tensor& qweight, const at::Tensor& meta, const c10::optional<at::Tensor>& bias,
                  int64_t bits, int64_t mode, int64_t N, int64_t K) {
    need_lib();
    check_x(x, K);
    const int group = check_native(qweight, meta, bits, N, K, x);
    const int64_t M = x.numel() / K;
    TORCH_CHECK(M >= 1 && M <= 8, "linear: 1..8 rows (got ", M, "); more rows take ops.linear / ops.gemm");
    if (bias) TORCH_CHECK(bias->scalar_type() == at::kHalf && bias->numel() == N && bias->device() == x.device(), "bias: fp16 [N] on x's device");
    auto sizes = x.sizes().vec();
    sizes.back() = N;
    at::Tensor y = at::empty(sizes, x.options());
    const c10::hip::HIPGuardMasqueradingAsCUDA dev_guard_(x.device());      // the tensors' device is current for the launch (attributes, CU count, null stream)
    check_rc(g_linear((int)bits, (int)mode, x.data_ptr(), qweight.data_ptr(), meta.data_ptr(), bias ? bias->data_ptr() : nullptr,
                      y.data_ptr(), (int)M, (int)N, (int)K, group, stream_of(x)), "amq_linear_f16");
    return y;
}

// several linears over the same x as segments of ONE launch (q/k/v, gate/up): amq_gemv_grouped_f16.  prologue 0 none,
// 1 RMSNorm (aux = gamma [K], eps), 2 SiLU*mul (aux = up [.., K]).  Returns one output tensor per segment.
std::vector<at::Tensor> grouped(const at::Tensor& x, const std::vector<at::Tensor>& qweights, const std::vector<at::Tensor>& metas,
                                const std::vector<int64_t>& Ns, const std::vector<int64_t>& bits, const std::vector<int64_t>& modes,
                                int64_t K, int64_t prologue, const c10::optional<at::Tensor>& aux, double eps) {
    need_lib();
    check_x(x, K);
    const size_t n = qweights.size();
    TORCH_CHECK(n >= 1 && n <= 4 && metas.size() == n && Ns.size() == n && bits.size() == n && modes.size() == n, "1..4 segments");
    const int64_t M = x.numel() / K;
    TORCH_CHECK(M >= 1 && M <= 16, "grouped: 1..16 rows (got ", M, ")");
    Segment segs[4];
    std::vector<at::Tensor> ys;
    ys.reserve(n);
    auto sizes = x.sizes().vec();
    int group = 0;
    for (size_t i = 0; i < n; ++i) {
        const int g = check_native(qweights[i], metas[i], bits[i], Ns[i], K, x);
        TORCH_CHECK(group == 0 || g == group, "segments of one launch must share their group size");
        group = g;
        sizes.back() = Ns[i];
        ys.push_back(at::empty(sizes, x.options()));
        segs[i] = Segment{qweights[i].data_ptr(), metas[i].data_ptr(), nullptr, nullptr, ys[i].data_ptr(), (int)Ns[i], (int)bits[i], (int)modes[i], 0};
    }
    const void* x2 = nullptr;
    const void* gamma = nullptr;
    if (prologue == 1) {
        TORCH_CHECK(aux && aux->scalar_type() == at::kHalf && aux->numel() == K && aux->device() == x.device(), "RMSNorm prologue: gamma fp16 [K]");
        gamma = aux->data_ptr();
    } else if (prologue == 2) {
        TORCH_CHECK(aux && aux->scalar_type() == at::kHalf && aux->numel() == x.numel() && aux->is_contiguous() && aux->device() == x.device(),
                    "SiLU*mul prologue: up fp16 of x's shape");
        x2 = aux->data_ptr();
    } else {
        TORCH_CHECK(prologue == 0, "unknown prologue");
    }
    const c10::hip::HIPGuardMasqueradingAsCUDA dev_guard_(x.device());      // the tensors' device is current for the launch (attributes, CU count, null stream)
    check_rc(g_grouped(segs, (int)n, x.data_ptr(), x2, gamma, (float)eps, (int)prologue, (int)M, (int)K, group, 0, nullptr, stream_of(x)),
             "amq_gemv_grouped_f16");
    return ys;
}

// the same launch with everything about the WEIGHTS checked once: a module keeps the handle (LinearGroup, HIPLlamaMLP) and a forward
// converts one tensor instead of five Python lists.  The handle holds the weight tensors, so their storage cannot go away under it;
// the owner drops the handle when its buffers are replaced (.to(), load_state_dict).
struct Group {
    std::vector<at::Tensor> qw, mt;
    std::vector<Segment> segs;
    int64_t K;
    int group = 0;              // granularity of the members' native meta (all members alike): 128, 64 or 32

    Group(const std::vector<at::Tensor>& qweights, const std::vector<at::Tensor>& metas, const std::vector<int64_t>& Ns,
          const std::vector<int64_t>& bits, const std::vector<int64_t>& modes, int64_t K_) : qw(qweights), mt(metas), K(K_) {
        const size_t n = qw.size();
        TORCH_CHECK(n >= 1 && n <= 4 && mt.size() == n && Ns.size() == n && bits.size() == n && modes.size() == n, "1..4 segments");
        for (size_t i = 0; i < n; ++i) {
            TORCH_CHECK(qw[i].is_cuda() && qw[i].device() == qw[0].device(), "the weights of a group live on one GPU");
            const int g = check_native(qw[i], mt[i], bits[i], Ns[i], K, qw[0]);
            TORCH_CHECK(group == 0 || g == group, "the members of a group must share their group size");
            group = g;
            segs.push_back(Segment{qw[i].data_ptr(), mt[i].data_ptr(), nullptr, nullptr, nullptr, (int)Ns[i], (int)bits[i], (int)modes[i], 0});
        }
    }

    // residual (single-member groups: o_proj, down_proj): y = residual + fp16(x . W^T), the runner's epilogue form
    std::vector<at::Tensor> run(const at::Tensor& x, int64_t prologue, const c10::optional<at::Tensor>& aux, double eps,
                                const c10::optional<at::Tensor>& residual) const {
        need_lib();
        check_x(x, K);
        TORCH_CHECK(x.device() == qw[0].device(), "x and the weights must be on the same device");
        const int64_t M = x.numel() / K;
        TORCH_CHECK(M >= 1 && M <= 16, "grouped: 1..16 rows (got ", M, ")");
        Segment local[4];
        std::vector<at::Tensor> ys;
        ys.reserve(segs.size());
        auto sizes = x.sizes().vec();
        for (size_t i = 0; i < segs.size(); ++i) {
            sizes.back() = segs[i].N;
            ys.push_back(at::empty(sizes, x.options()));
            local[i] = segs[i];
            local[i].y = ys[i].data_ptr();
        }
        if (residual) {
            TORCH_CHECK(segs.size() == 1 && residual->scalar_type() == at::kHalf && residual->is_contiguous() && residual->device() == x.device() &&
                            residual->numel() == M * segs[0].N,
                        "residual: a contiguous fp16 tensor of the (single) output's shape");
            local[0].residual = residual->data_ptr();
        }
        const void* x2 = nullptr;
        const void* gamma = nullptr;
        if (prologue == 1) {
            TORCH_CHECK(aux && aux->scalar_type() == at::kHalf && aux->numel() == K && aux->device() == x.device(), "RMSNorm prologue: gamma fp16 [K]");
            gamma = aux->data_ptr();
        } else if (prologue == 2) {
            TORCH_CHECK(aux && aux->scalar_type() == at::kHalf && aux->numel() == x.numel() && aux->is_contiguous() && aux->device() == x.device(),
                        "SiLU*mul prologue: up fp16 of x's shape");
            x2 = aux->data_ptr();
        } else {
            TORCH_CHECK(prologue == 0, "unknown prologue");
        }
        const c10::hip::HIPGuardMasqueradingAsCUDA dev_guard_(x.device());      // the tensors' device is current for the launch (attributes, CU count, null stream)
        check_rc(g_grouped(local, (int)segs.size(), x.data_ptr(), x2, gamma, (float)eps, (int)prologue, (int)M, (int)K, group, 0, nullptr, stream_of(x)),
                 "amq_gemv_grouped_f16");
        return ys;
    }
};

// one decode step of attention over the step-state block's cos/sin row (amq_attn_decode_cur_f16): rotates q / k, appends k / v to
// the caches at the block's position, returns softmax(q K^T / sqrt(128)) V as a new tensor of q's shape
at::Tensor attn_decode_cur(const at::Tensor& q, const at::Tensor& k, const at::Tensor& v, const at::Tensor& kcache, const at::Tensor& vcache,
                           const at::Tensor& cur, int64_t n_heads, int64_t n_kv_heads) {
    need_lib();
    TORCH_CHECK(kcache.dim() == 4 && kcache.size(1) == n_kv_heads && kcache.size(3) == 128 && vcache.sizes() == kcache.sizes(),
                "caches: [B, n_kv_heads, max_seq, 128]");
    const int64_t B = kcache.size(0), max_seq = kcache.size(2);
    for (const at::Tensor* t : {&q, &k, &v, &kcache, &vcache, &cur})
        TORCH_CHECK(t->is_cuda() && t->scalar_type() == at::kHalf && t->is_contiguous() && t->device() == q.device(), "attn_decode_cur: contiguous fp16 tensors on one GPU");
    TORCH_CHECK(q.numel() == B * n_heads * 128 && k.numel() == B * n_kv_heads * 128 && v.numel() == k.numel() && cur.numel() == 128,
                "attn_decode_cur: q [B, n_heads*128], k / v [B, n_kv_heads*128], cur [128]");
    at::Tensor out = at::empty_like(q);
    const c10::hip::HIPGuardMasqueradingAsCUDA dev_guard_(q.device());      // the tensors' device is current for the launch (attributes, CU count, null stream)
    check_rc(g_attn_cur(q.data_ptr(), k.data_ptr(), v.data_ptr(), kcache.data_ptr(), vcache.data_ptr(), out.data_ptr(), cur.data_ptr(), (int)B,
                        (int)n_heads, (int)n_kv_heads, 128, (int)max_seq, stream_of(q)), "amq_attn_decode_cur_f16");
    return out;
}

at::Tensor rmsnorm(const at::Tensor& x, const at::Tensor& gamma, double eps) {
    need_lib();
    const int64_t K = x.size(-1);
    check_x(x, K);
    TORCH_CHECK(gamma.scalar_type() == at::kHalf && gamma.numel() == K && gamma.device() == x.device(), "gamma: fp16 [K] on x's device");
    at::Tensor y = at::empty_like(x);
    const c10::hip::HIPGuardMasqueradingAsCUDA dev_guard_(x.device());      // the tensors' device is current for the launch (attributes, CU count, null stream)
    check_rc(g_rmsnorm(x.data_ptr(), gamma.data_ptr(), y.data_ptr(), (int)(x.numel() / K), (int)K, (float)eps, stream_of(x)), "amq_rmsnorm_f16");
    return y;
}

at::Tensor silu_mul(const at::Tensor& gate, const at::Tensor& up) {
    need_lib();
    TORCH_CHECK(gate.is_cuda() && gate.scalar_type() == at::kHalf && gate.is_contiguous() && up.scalar_type() == at::kHalf && up.is_contiguous() &&
                    up.numel() == gate.numel() && up.device() == gate.device() && gate.numel() % 8 == 0,
                "silu_mul: two contiguous fp16 tensors of one size (a multiple of 8) on one device");
    at::Tensor y = at::empty_like(gate);
    const c10::hip::HIPGuardMasqueradingAsCUDA dev_guard_(gate.device());      // the tensors' device is current for the launch (attributes, CU count, null stream)
    check_rc(g_silu_mul(gate.data_ptr(), up.data_ptr(), y.data_ptr(), (size_t)gate.numel(), stream_of(gate)), "amq_silu_mul_f16");
    return y;
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.def("init", &init, "dlopen libamq_hip.so and resolve the entry points");
    m.def("linear", &linear, "y = x . W^T (+ bias), 1..8 rows");
    m.def("grouped", &grouped, "several linears over one x in one launch");
    pybind11::class_<Group, std::shared_ptr<Group>>(m, "Group")
        .def(pybind11::init<const std::vector<at::Tensor>&, const std::vector<at::Tensor>&, const std::vector<int64_t>&, const std::vector<int64_t>&,
                            const std::vector<int64_t>&, int64_t>())
        .def("run", &Group::run, pybind11::arg("x"), pybind11::arg("prologue"), pybind11::arg("aux"), pybind11::arg("eps"),
             pybind11::arg("residual") = pybind11::none(), "the group's launch over x: (x, prologue, aux, eps[, residual]) -> one output per member");
    m.def("attn_decode_cur", &attn_decode_cur);
    m.def("rmsnorm", &rmsnorm);
    m.def("silu_mul", &silu_mul);
}
