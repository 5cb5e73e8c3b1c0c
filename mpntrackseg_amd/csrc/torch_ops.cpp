// torch.ops.mpnhip.* -- the hot path's operators registered with the PyTorch dispatcher (TORCH_LIBRARY), as SURVEY.md
// section 8b words the boundary ("PyTorch custom ops registered with TORCH_LIBRARY(mpnhip, ...)").  A thin C++ shim over the
// C ABI of include/mpnhip.h: no kernel lives here, nothing is hipified, every op allocates its outputs with torch's caching
// allocator on the inputs' device and launches on c10::hip::getCurrentHIPStream() -- exactly what the ctypes binding
// (mpntrackseg_amd/capi.py) does from Python, minus the per-call struct marshalling in the interpreter.
//
// The model (reference MOTMPNet.__init__, models/mpn.py:220-317) crosses the op boundary as
//   spec    int[]     {dn, de, reattach_nodes, reattach_edges, agg, num_enc_steps, precision,
//                      then for each of enc_node, enc_edge, edge, flow_in, flow_out, node, classifier: n_layers, in_dim, out_dims...}
//   weights Tensor[]  {w0, b0, w1, b1, ...} of the seven MLPs in that order (nn.Linear layout, fp32, contiguous)
// Built by the Makefile with g++ against the torch headers (torch_ops target) into csrc/libmpnhip_torch.so.
#include <ATen/ATen.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>   // (PyTorch-ROCm presents HIP devices under the "cuda" device type)
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <torch/library.h>

#include <vector>

#include "../../include/mpnhip.h"

namespace {

void check_rc(int rc, const char* what) { TORCH_CHECK(rc == MPNHIP_OK, what, " failed (code ", rc, "): ", mpnhip_last_error()); }

void* cur_stream() { return static_cast<void*>(c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream()); }

const float* fptr(const at::Tensor& t) { return t.defined() && t.numel() ? t.data_ptr<float>() : nullptr; }

void check_f32(const at::Tensor& t, const char* name) {
    TORCH_CHECK(t.is_cuda() && t.scalar_type() == at::kFloat && t.is_contiguous(), "mpnhip: ", name, " must be a contiguous float32 HIP tensor");
}

// spec / weights -> mpnhip_model (pointers borrowed from `weights`; grads: optional parallel list, same order)
mpnhip_model make_model(at::IntArrayRef spec, at::TensorList weights, const std::vector<at::Tensor>* grads) {
    TORCH_CHECK(spec.size() >= 7, "mpnhip: model spec too short");
    mpnhip_model m = {};
    m.dn = (int)spec[0]; m.de = (int)spec[1]; m.reattach_nodes = (int)spec[2]; m.reattach_edges = (int)spec[3];
    m.agg = (int)spec[4]; m.num_enc_steps = (int)spec[5]; m.precision = (int)spec[6];
    mpnhip_mlp* mlps[7] = {&m.enc_node, &m.enc_edge, &m.edge, &m.flow_in, &m.flow_out, &m.node, &m.classifier};
    size_t si = 7, wi = 0;
    for (mpnhip_mlp* q : mlps) {
        TORCH_CHECK(si + 2 <= spec.size(), "mpnhip: model spec truncated");
        q->n_layers = (int)spec[si++];
        q->in_dim = (int)spec[si++];
        TORCH_CHECK(q->n_layers >= 1 && q->n_layers <= MPNHIP_MAX_LAYERS && si + q->n_layers <= spec.size(), "mpnhip: bad MLP depth in spec");
        for (int i = 0; i < q->n_layers; ++i) {
            q->out_dims[i] = (int)spec[si++];
            TORCH_CHECK(wi + 2 <= weights.size(), "mpnhip: too few weight tensors");
            const at::Tensor& w = weights[wi];
            const at::Tensor& b = weights[wi + 1];
            check_f32(w, "weight");
            check_f32(b, "bias");
            const int in = i == 0 ? q->in_dim : q->out_dims[i - 1];
            TORCH_CHECK(w.dim() == 2 && w.size(0) == q->out_dims[i] && w.size(1) == in && b.numel() == q->out_dims[i],
                        "mpnhip: weight shape does not match the spec");
            q->weight[i] = w.data_ptr<float>();
            q->bias[i] = b.data_ptr<float>();
            if (grads) {
                q->grad_weight[i] = (*grads)[wi].data_ptr<float>();
                q->grad_bias[i] = (*grads)[wi + 1].data_ptr<float>();
            }
            wi += 2;
        }
    }
    TORCH_CHECK(wi == weights.size() && si == spec.size(), "mpnhip: spec and weight list disagree");
    return m;
}

at::Tensor bytes_like(const at::Tensor& ref, size_t n) {
    return at::empty({(int64_t)(n > 256 ? n : 256)}, ref.options().dtype(at::kByte));
}

// mpnhip_graph_prep (replaces the boolean-mask indexing of models/mpn.py:85-93): opaque uint8 graph buffer
at::Tensor graph_prep(const at::Tensor& edge_index, int64_t n_nodes, bool full) {
    TORCH_CHECK(edge_index.is_cuda() && edge_index.scalar_type() == at::kLong && edge_index.dim() == 2 && edge_index.size(0) == 2,
                "mpnhip::graph_prep: edge_index must be an int64 [2, E] HIP tensor (reference data/mot_graph.py:312)");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(edge_index.device());
    const at::Tensor ei = edge_index.contiguous();
    const int64_t E = ei.size(1);
    at::Tensor buf = bytes_like(ei, mpnhip_graph_bytes((int)n_nodes, E));
    at::Tensor ws = bytes_like(ei, mpnhip_graph_prep_workspace_bytes((int)n_nodes, E));
    auto fn = full ? mpnhip_graph_prep : mpnhip_graph_prep_forward;
    check_rc(fn(ei.data_ptr<int64_t>(), (int)n_nodes, E, buf.data_ptr(), (size_t)buf.numel(), ws.data_ptr(), (size_t)ws.numel(), cur_stream()),
             "mpnhip_graph_prep");
    return buf;
}

// mpnhip_forward (MOTMPNet.forward hot path, models/mpn.py:349-392): (logits [max(L,1), E], workspace)
// save != 0: the workspace holds every step's activations for mpnhip::backward
// workspace (optional): a buffer to run in instead of a fresh one (reused when large enough); weights_prepacked: the caller vouches
// that it still holds THIS model's packed weight images from an earlier inference call (mpnhip_model.weights_prepacked)
std::tuple<at::Tensor, at::Tensor> forward(const at::Tensor& graph, const at::Tensor& x, const at::Tensor& edge_attr, at::TensorList weights,
                                           at::IntArrayRef spec, int64_t save, const c10::optional<at::Tensor>& workspace,
                                           bool weights_prepacked) {
    check_f32(x, "x");
    check_f32(edge_attr, "edge_attr");
    TORCH_CHECK(x.dim() == 2 && edge_attr.dim() == 2, "mpnhip::forward: x [N, node_in_dim], edge_attr [E, edge_in_dim]");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(x.device());
    mpnhip_model m = make_model(spec, weights, nullptr);
    TORCH_CHECK(x.size(1) == m.enc_node.in_dim && edge_attr.size(1) == m.enc_edge.in_dim, "mpnhip::forward: input feature widths do not match the encoder");
    const int64_t N = x.size(0), E = edge_attr.size(0);
    TORCH_CHECK((size_t)graph.numel() >= mpnhip_graph_bytes((int)N, E), "mpnhip::forward: the prepared graph does not describe N nodes / E edges");
    const int64_t L = m.num_enc_steps > 0 ? m.num_enc_steps : 1;
    at::Tensor logits = at::empty({L, E}, x.options());
    const size_t need = mpnhip_forward_workspace_bytes(&m, (int)N, E, (int)save);
    at::Tensor ws;
    if (workspace.has_value() && workspace->defined() && workspace->is_cuda() && (size_t)workspace->numel() >= need && workspace->is_contiguous()) {
        ws = *workspace;
        m.weights_prepacked = weights_prepacked && !save ? 1 : 0;
    } else {
        ws = bytes_like(x, need);
    }
    check_rc(mpnhip_forward(&m, graph.data_ptr(), (int)N, E, fptr(x), fptr(edge_attr), logits.data_ptr<float>(), nullptr, nullptr, ws.data_ptr(),
                            (size_t)ws.numel(), (int)save, cur_stream()), "mpnhip_forward");
    return std::make_tuple(logits, ws);
}

// mpnhip_backward: returns {grad of every weight tensor in list order..., grad_x, grad_edge_attr} (the last two empty if not asked)
std::vector<at::Tensor> backward(const at::Tensor& graph, const at::Tensor& x, const at::Tensor& edge_attr, const at::Tensor& grad_logits,
                                 const at::Tensor& fwd_workspace, at::TensorList weights, at::IntArrayRef spec, bool need_gx, bool need_gea) {
    check_f32(x, "x");
    check_f32(edge_attr, "edge_attr");
    check_f32(grad_logits, "grad_logits");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(x.device());
    std::vector<at::Tensor> grads;
    grads.reserve(weights.size() + 2);
    for (const at::Tensor& w : weights) grads.push_back(at::zeros_like(w));
    mpnhip_model m = make_model(spec, weights, &grads);
    const int64_t N = x.size(0), E = edge_attr.size(0);
    at::Tensor gx = need_gx ? at::empty_like(x) : at::Tensor();
    at::Tensor gea = need_gea ? at::empty_like(edge_attr) : at::Tensor();
    at::Tensor bws = bytes_like(x, mpnhip_backward_workspace_bytes(&m, (int)N, E));
    check_rc(mpnhip_backward(&m, graph.data_ptr(), (int)N, E, fptr(x), fptr(edge_attr), fptr(grad_logits), nullptr, nullptr,
                             need_gx ? gx.data_ptr<float>() : nullptr, need_gea ? gea.data_ptr<float>() : nullptr, fwd_workspace.data_ptr(),
                             (size_t)fwd_workspace.numel(), bws.data_ptr(), (size_t)bws.numel(), cur_stream()), "mpnhip_backward");
    grads.push_back(need_gx ? gx : at::empty({0}, x.options()));
    grads.push_back(need_gea ? gea : at::empty({0}, x.options()));
    return grads;
}

// mpnhip_meta_layer_forward (MetaLayer.forward, models/mpn.py:33-54): weights / spec as above (only the MetaLayer parts are read)
std::tuple<at::Tensor, at::Tensor> meta_layer(const at::Tensor& graph, const at::Tensor& x, const at::Tensor& e, at::TensorList weights,
                                              at::IntArrayRef spec) {
    check_f32(x, "x");
    check_f32(e, "edge_attr");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(x.device());
    mpnhip_model m = make_model(spec, weights, nullptr);
    const int64_t N = x.size(0), E = e.size(0);
    TORCH_CHECK((x.size(1) == m.dn || x.size(1) == 2 * m.dn) && (e.size(1) == m.de || e.size(1) == 2 * m.de),
                "mpnhip::meta_layer: input widths do not match the edge / flow MLP dims");
    m.reattach_nodes = x.size(1) == 2 * m.dn;
    m.reattach_edges = e.size(1) == 2 * m.de;
    at::Tensor x_new = at::empty({N, m.dn}, x.options());
    at::Tensor e_new = at::empty({E, m.de}, x.options());
    at::Tensor ws = bytes_like(x, mpnhip_meta_layer_workspace_bytes(&m, (int)N, E));
    check_rc(mpnhip_meta_layer_forward(&m, graph.data_ptr(), (int)N, E, fptr(x), fptr(e), x_new.data_ptr<float>(), e_new.data_ptr<float>(),
                                       ws.data_ptr(), (size_t)ws.numel(), cur_stream()), "mpnhip_meta_layer_forward");
    return std::make_tuple(x_new, e_new);
}

// node_agg_fn(out, row, x_size) (models/mpn.py:266-273)
at::Tensor segment_reduce(const at::Tensor& src, const at::Tensor& row, int64_t x_size, int64_t agg) {
    check_f32(src, "src");
    TORCH_CHECK(row.is_cuda() && row.scalar_type() == at::kLong && row.dim() == 1 && row.size(0) == src.size(0), "mpnhip::segment_reduce: row must be int64 [M]");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(src.device());
    const int64_t M = src.size(0);
    const int64_t dim = M ? src.numel() / M : (src.dim() > 1 ? src.size(1) : 1);
    std::vector<int64_t> shape(src.sizes().begin(), src.sizes().end());
    shape[0] = x_size;
    at::Tensor out = at::empty(shape, src.options());
    at::Tensor ws = bytes_like(src, mpnhip_segment_reduce_workspace_bytes(M, (int)x_size));
    const at::Tensor r = row.contiguous();
    check_rc(mpnhip_segment_reduce(fptr(src), r.data_ptr<int64_t>(), M, (int)dim, (int)x_size, (int)agg, out.data_ptr<float>(), nullptr, ws.data_ptr(),
                                   (size_t)ws.numel(), cur_stream()), "mpnhip_segment_reduce");
    return out;
}

}  // namespace

TORCH_LIBRARY(mpnhip, m) {
    m.def("graph_prep(Tensor edge_index, int n_nodes, bool full=True) -> Tensor");
    m.def("forward(Tensor graph, Tensor x, Tensor edge_attr, Tensor[] weights, int[] spec, int save=0, Tensor? workspace=None, "
          "bool weights_prepacked=False) -> (Tensor, Tensor)");
    m.def("backward(Tensor graph, Tensor x, Tensor edge_attr, Tensor grad_logits, Tensor fwd_workspace, Tensor[] weights, int[] spec, "
          "bool need_grad_x=False, bool need_grad_edge_attr=False) -> Tensor[]");
    m.def("meta_layer(Tensor graph, Tensor x, Tensor edge_attr, Tensor[] weights, int[] spec) -> (Tensor, Tensor)");
    m.def("segment_reduce(Tensor src, Tensor row, int x_size, int agg) -> Tensor");
}

TORCH_LIBRARY_IMPL(mpnhip, CUDA, m) {   // ("CUDA" is the dispatch key of HIP devices on PyTorch-ROCm)
    m.impl("graph_prep", graph_prep);
    m.impl("forward", forward);
    m.impl("backward", backward);
    m.impl("meta_layer", meta_layer);
    m.impl("segment_reduce", segment_reduce);
}
