// Context, memory helpers, layer-program executor (pp_net_*) and the single-conv entry point.
#include <dlfcn.h>

#include <algorithm>
#include <memory>

#include "pp_internal.h"
#include "pp_amax.h"

#include <atomic>
extern std::atomic<int> g_decode_generic, g_split_gemm_epi;      // dark_decode.hip, conv_split.hip

static thread_local std::string g_last_error;

void pp_set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

namespace {
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    Roctx() {
        for (const char* lib : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
            void* h = dlopen(lib, RTLD_LAZY | RTLD_LOCAL);
            if (!h) continue;
            push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
            pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
            if (push && pop) return;
            push = nullptr; pop = nullptr;
        }
    }
};
const Roctx& roctx() {
    static const Roctx r;
    return r;
}
}  // namespace

void pp_range_push(const char* name) {
    if (roctx().push) (void)roctx().push(name);
}
void pp_range_pop() {
    if (roctx().pop) (void)roctx().pop();
}

int pp_ctx::ensure_scratch(size_t bytes) {
    if (bytes <= scratch_bytes) return PP_OK;
    if (scratch) {
        PP_HIP_CHECK(hipStreamSynchronize(stream));
        PP_HIP_CHECK(hipFree(scratch));
        scratch = nullptr;
        scratch_bytes = 0;
    }
    size_t want = std::max(bytes, size_t(1) << 20);
    PP_HIP_CHECK(hipMalloc(&scratch, want));
    scratch_bytes = want;
    return PP_OK;
}

extern "C" {

int pp_abi_version(void) { return PP_ABI_VERSION; }

int pp_debug_knob(const char* name, int value) {
    PP_REQUIRE(name && value >= -1 && value <= 1, "pp_debug_knob: name and a value in {-1, 0, 1}");
    if (!strcmp(name, "decode_generic")) g_decode_generic.store(value, std::memory_order_relaxed);
    else if (!strcmp(name, "split_gemm_epilogue")) g_split_gemm_epi.store(value, std::memory_order_relaxed);
    else {
        pp_set_error("pp_debug_knob: unknown knob '%s' (decode_generic, split_gemm_epilogue)", name);
        return PP_ERR_ARG;
    }
    return PP_OK;
}

const char* pp_last_error(void) { return g_last_error.c_str(); }

int pp_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

int pp_ctx_create(int device, pp_ctx** out) {
    PP_REQUIRE(out != nullptr, "pp_ctx_create: out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        pp_set_error("pp_ctx_create: no HIP device visible (%s) -- this library has no CPU fallback",
                     e == hipSuccess ? "device count 0" : hipGetErrorString(e));
        return PP_ERR_HIP;
    }
    PP_REQUIRE(device >= 0 && device < n, "pp_ctx_create: device %d out of range [0,%d)", device, n);
    PP_HIP_CHECK(hipSetDevice(device));
    std::unique_ptr<pp_ctx> c(new pp_ctx());
    c->device = device;
    PP_HIP_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    PP_HIP_CHECK(hipEventCreate(&c->ev_start));
    PP_HIP_CHECK(hipEventCreate(&c->ev_stop));
    PP_HIP_CHECK(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    PP_HIP_CHECK(hipEventCreateWithFlags(&c->ev_upload, hipEventDisableTiming));
    PP_HIP_CHECK(hipEventCreateWithFlags(&c->ev_consumed, hipEventDisableTiming));
    *out = c.release();
    return PP_OK;
}

void pp_ctx_destroy(pp_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    if (ctx->ev_start) (void)hipEventDestroy(ctx->ev_start);
    if (ctx->ev_stop) (void)hipEventDestroy(ctx->ev_stop);
    if (ctx->copy_stream) {
        (void)hipStreamSynchronize(ctx->copy_stream);
        (void)hipStreamDestroy(ctx->copy_stream);
    }
    if (ctx->ev_upload) (void)hipEventDestroy(ctx->ev_upload);
    if (ctx->ev_consumed) (void)hipEventDestroy(ctx->ev_consumed);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int pp_ctx_set_stream(pp_ctx* ctx, void* hip_stream) {
    PP_REQUIRE(ctx, "pp_ctx_set_stream: ctx is NULL");
    PP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (ctx->own_stream && ctx->stream) PP_HIP_CHECK(hipStreamDestroy(ctx->stream));
    ctx->stream = static_cast<hipStream_t>(hip_stream);
    ctx->own_stream = false;
    return PP_OK;
}

int pp_ctx_synchronize(pp_ctx* ctx) {
    PP_REQUIRE(ctx, "pp_ctx_synchronize: ctx is NULL");
    PP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return PP_OK;
}

int pp_timer_start(pp_ctx* ctx) {
    PP_REQUIRE(ctx, "pp_timer_start: ctx is NULL");
    PP_HIP_CHECK(hipEventRecord(ctx->ev_start, ctx->stream));
    return PP_OK;
}

int pp_timer_stop(pp_ctx* ctx, float* elapsed_ms) {
    PP_REQUIRE(ctx && elapsed_ms, "pp_timer_stop: NULL argument");
    PP_HIP_CHECK(hipEventRecord(ctx->ev_stop, ctx->stream));
    PP_HIP_CHECK(hipEventSynchronize(ctx->ev_stop));
    PP_HIP_CHECK(hipEventElapsedTime(elapsed_ms, ctx->ev_start, ctx->ev_stop));
    return PP_OK;
}

int pp_malloc(pp_ctx* ctx, size_t bytes, void** dptr) {
    PP_REQUIRE(ctx && dptr, "pp_malloc: NULL argument");
    PP_HIP_CHECK(hipSetDevice(ctx->device));
    PP_HIP_CHECK(hipMalloc(dptr, bytes ? bytes : 1));
    return PP_OK;
}

int pp_free(pp_ctx* ctx, void* dptr) {
    PP_REQUIRE(ctx, "pp_free: ctx is NULL");
    if (!dptr) return PP_OK;
    PP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    PP_HIP_CHECK(hipFree(dptr));
    return PP_OK;
}

int pp_memcpy_h2d(pp_ctx* ctx, void* dst, const void* src, size_t bytes) {
    PP_REQUIRE(ctx && (bytes == 0 || (dst && src)), "pp_memcpy_h2d: NULL argument");
    PP_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    PP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return PP_OK;
}

int pp_host_alloc(pp_ctx* ctx, size_t bytes, void** out) {
    PP_REQUIRE(ctx && out, "pp_host_alloc: NULL argument");
    *out = nullptr;
    PP_HIP_CHECK(hipSetDevice(ctx->device));
    PP_HIP_CHECK(hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
    return PP_OK;
}

int pp_host_free(pp_ctx* ctx, void* p) {
    PP_REQUIRE(ctx != nullptr, "pp_host_free: ctx is NULL");
    if (p) PP_HIP_CHECK(hipHostFree(p));
    return PP_OK;
}

int pp_upload_begin(pp_ctx* ctx, void* dst, const void* src, size_t bytes) {
    PP_REQUIRE(ctx && (bytes == 0 || (dst && src)), "pp_upload_begin: NULL argument");
    // the destination may still be read by compute work enqueued before the last pp_upload_release
    PP_HIP_CHECK(hipStreamWaitEvent(ctx->copy_stream, ctx->ev_consumed, 0));
    PP_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->copy_stream));
    PP_HIP_CHECK(hipEventRecord(ctx->ev_upload, ctx->copy_stream));
    return PP_OK;
}

int pp_upload_begin_nv12(pp_ctx* ctx, void* dst_bgr, void* tmp_nv12, const void* src_host, int frames, int height, int width) {
    PP_REQUIRE(ctx && (frames == 0 || (dst_bgr && tmp_nv12 && src_host)), "pp_upload_begin_nv12: NULL argument");
    PP_REQUIRE(frames >= 0 && height > 0 && width > 0, "pp_upload_begin_nv12: bad shape");
    PP_HIP_CHECK(hipStreamWaitEvent(ctx->copy_stream, ctx->ev_consumed, 0));
    const size_t bytes = (size_t)frames * height * width * 3 / 2;
    if (bytes) PP_HIP_CHECK(hipMemcpyAsync(tmp_nv12, src_host, bytes, hipMemcpyHostToDevice, ctx->copy_stream));
    // the conversion rides the copy stream: `ev_upload` then covers the BGR frames, and the (in-order) copy stream does not
    // overwrite tmp_nv12 with the next chunk before this kernel has read it
    const int r = pp_launch_nv12_to_bgr((const unsigned char*)tmp_nv12, (unsigned char*)dst_bgr, frames, height, width, ctx->copy_stream);
    if (r != PP_OK) return r;
    PP_HIP_CHECK(hipEventRecord(ctx->ev_upload, ctx->copy_stream));
    return PP_OK;
}

int pp_nv12_to_bgr(pp_ctx* ctx, const uint8_t* nv12, int frames, int height, int width, uint8_t* bgr) {
    PP_REQUIRE(ctx && (frames == 0 || (nv12 && bgr)), "pp_nv12_to_bgr: NULL argument");
    return pp_launch_nv12_to_bgr(nv12, bgr, frames, height, width, ctx->stream);
}

int pp_upload_wait(pp_ctx* ctx, int host_sync) {
    PP_REQUIRE(ctx != nullptr, "pp_upload_wait: ctx is NULL");
    PP_HIP_CHECK(hipStreamWaitEvent(ctx->stream, ctx->ev_upload, 0));
    if (host_sync) PP_HIP_CHECK(hipEventSynchronize(ctx->ev_upload));
    return PP_OK;
}

int pp_upload_release(pp_ctx* ctx) {
    PP_REQUIRE(ctx != nullptr, "pp_upload_release: ctx is NULL");
    PP_HIP_CHECK(hipEventRecord(ctx->ev_consumed, ctx->stream));
    return PP_OK;
}

int pp_memcpy_d2h(pp_ctx* ctx, void* dst, const void* src, size_t bytes) {
    PP_REQUIRE(ctx && (bytes == 0 || (dst && src)), "pp_memcpy_d2h: NULL argument");
    PP_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    PP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return PP_OK;
}

int pp_memcpy_d2d(pp_ctx* ctx, void* dst, const void* src, size_t bytes) {
    PP_REQUIRE(ctx && (bytes == 0 || (dst && src)), "pp_memcpy_d2d: NULL argument");
    PP_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return PP_OK;
}

}  // extern "C"

// ---- layer programs ----------------------------------------------------------------------------

struct pp_net {
    pp_ctx* ctx = nullptr;
    std::vector<pp_op> ops;
    std::vector<pp_buf> bufs;
    std::vector<size_t> buf_off;    // float offset of buffer b (sized for max_batch)
    std::vector<size_t> buf_elems;  // per-sample floats
    float* weights = nullptr;
    size_t n_weights = 0;
    // conv_split.hip: the eligible convs' weights as bf16 planes in fragment order (built on the device at creation)
    unsigned char* wsplit = nullptr;
    std::vector<long long> wsplit_off;   // per op: byte offset into wsplit, -1: the op runs on the fp32-MFMA kernels
    int numerics = PP_NET_NUMERICS_EXACT;   // fixed at creation (ABI 7)
    int split_f16 = 0;                      // split nets: the fp16 form (ABI 9), fixed at creation as well
    // fp16 form (pp_amax.h): per-sample running maxima of the tensors its convolutions read, [slot][max_batch] bit patterns.  A slot
    // belongs to ONE tensor of the program (a buffer between two overwrites); it is raised by the fused epilogues of the tensor's
    // producers (op_y_slot) or, where a producer has none, by a stand-alone pass after the last of them (op_post_slot); tensors that
    // come from outside the program (`ext`) get one pass at the start of the run, unless their producer supplies the maxima itself
    // (pp_net_input_amax).  Tensor slots are numbered in op order, so the slots a run of ops [first, last) has to zero are one
    // contiguous range (slot_owner = the first op that raises the slot).
    unsigned* amax = nullptr;
    std::vector<int> op_x_slot, op_y_slot, op_post_slot, slot_owner;
    std::vector<int> slot_first, slot_last;      // tracked slots: first / last op that raises them (-1: nobody reads the maxima)
    struct ExtAmax { int buf, slot, first_reader, last_reader; bool provided; };
    std::vector<ExtAmax> ext;      // tensors from outside the program that fp16-form convolutions read
    float* arena = nullptr;
    size_t arena_floats = 0;
    int max_batch = 0;
    hipGraphExec_t graph_exec = nullptr;
    int graph_batch = 0;
    bool use_lanes = true;
    // multi-lane execution: independent ops (HRNet branches, FPN / RPN levels) run on separate HIP streams so
    // that the tail of one launch overlaps the head of another; dependencies (RAW on inputs / residuals, WAR
    // and WAW on recycled buffers) become event waits.
    std::vector<hipStream_t> lanes;
    std::vector<int> op_lane;
    std::vector<std::vector<int>> op_waits;   // ops on OTHER lanes this op must wait for
    std::vector<char> op_needs_event;
    std::vector<hipEvent_t> op_done;
    hipEvent_t fork_ev = nullptr;
    std::vector<hipEvent_t> join_ev;
    std::vector<pp_vit_encoder*> vits;   // per op: the encoder of a PP_OP_VIT_ENCODER, else null
    std::vector<pp_deconv_bf16*> deconvs; // per op: the object of a PP_OP_DECONV_BF16, else null

    float* buf_ptr(int b) const { return arena + buf_off[b]; }
    unsigned* amax_slot(int slot) const { return slot >= 0 ? amax + (size_t)slot * max_batch : nullptr; }
};

static ConvArgs net_conv_args(pp_net* net, const pp_op& op, int batch);
static int net_make_lanes(pp_net* net, int n_lanes);

// which tensors the fp16-form convolutions read, who produces them and who therefore tracks their maxima (see pp_net::amax)
static void net_plan_amax(pp_net* net) {
    const int n = (int)net->ops.size(), nb = (int)net->bufs.size();
    net->op_x_slot.assign(n, -1);
    net->op_y_slot.assign(n, -1);
    net->op_post_slot.assign(n, -1);
    net->slot_owner.clear();
    net->ext.clear();
    if (net->numerics != PP_NET_NUMERICS_SPLIT || !net->split_f16) return;
    auto is_h = [&](int i) { return net->ops[i].type == PP_OP_CONV && net->wsplit_off[i] >= 0; };
    // producers with a fused maximum: convolutions (pp_conv_tracks_amax) and PP_OP_UPSAMPLE_ADD
    auto tracks = [&](int i) {
        const pp_op& op = net->ops[i];
        if (op.type == PP_OP_UPSAMPLE_ADD) return true;
        return op.type == PP_OP_CONV && pp_conv_tracks_amax(net_conv_args(net, op, 1), is_h(i));
    };
    // pass 1: buffers a fp16-form convolution reads BEFORE any op of the program has written them = tensors from outside (the
    // program's inputs).  One slot per buffer, numbered first; filled by one stand-alone pass at the start of a run that
    // contains a reader -- or by the caller's own producer (pp_net_input_amax: the detector's RoIAlign)
    {
        std::vector<char> written(nb, 0);
        for (int i = 0; i < n; ++i) {
            const pp_op& op = net->ops[i];
            if (is_h(i) && !written[op.in]) {
                pp_net::ExtAmax* e = nullptr;
                for (auto& x : net->ext)
                    if (x.buf == op.in) e = &x;
                if (!e) {
                    net->ext.push_back({op.in, (int)net->slot_owner.size(), i, i, false});
                    net->slot_owner.push_back(-1);    // (never part of the zeroed range: handled per run, below)
                    e = &net->ext.back();
                }
                e->last_reader = i;
                net->op_x_slot[i] = e->slot;
            }
            written[op.out] = 1;
        }
    }
    struct Tensor { int slot; std::vector<int> producers; bool wanted; };
    std::vector<Tensor> tensors;
    std::vector<int> cur(nb, -1);                 // tensor currently held by buffer b (-1: written outside the program)
    std::vector<char> read_since(nb, 1);
    std::vector<std::pair<int, int>> readers;     // (H conv, tensor)
    for (int i = 0; i < n; ++i) {
        const pp_op& op = net->ops[i];
        if (is_h(i) && cur[op.in] >= 0) {
            tensors[cur[op.in]].wanted = true;
            readers.push_back({i, cur[op.in]});
        }
        for (int b : {op.in, op.res1, op.res2, op.in2, op.in3})
            if (b >= 0) read_since[b] = 1;
        // a write after a read (or the first write) starts a new tensor; writes without a read in between fill the same one (the
        // channel slices of a concatenation)
        if (cur[op.out] < 0 || read_since[op.out]) {
            cur[op.out] = (int)tensors.size();
            tensors.push_back({(int)net->slot_owner.size(), {}, false});
            net->slot_owner.push_back(i);         // (a slot per tensor, wanted or not: the numbering must follow the op order)
        }
        tensors[cur[op.out]].producers.push_back(i);
        read_since[op.out] = 0;
    }
    net->slot_first.assign(net->slot_owner.size(), -1);
    net->slot_last.assign(net->slot_owner.size(), -1);
    for (const Tensor& t : tensors) {
        if (!t.wanted) continue;
        net->slot_first[t.slot] = t.producers.front();
        net->slot_last[t.slot] = t.producers.back();
        bool fused = true;
        for (int p : t.producers) fused = fused && tracks(p);
        // the fused maxima cover what the producers WRITE: when their channel slices do not add up to the whole buffer, part of
        // what the readers see is older content (a dense-connection pattern, a buffer partly filled from outside) that only the
        // stand-alone pass over the whole buffer accounts for
        {
            int covered = 0;
            const int out_b = net->ops[t.producers.front()].out;
            for (int p : t.producers) covered += net->ops[p].type == PP_OP_CONV ? net->ops[p].cout : net->bufs[out_b].c;
            if (covered < net->bufs[out_b].c) fused = false;
        }
        if (fused)
            for (int p : t.producers) net->op_y_slot[p] = t.slot;
        else
            net->op_post_slot[t.producers.back()] = t.slot;
    }
    for (const auto& r : readers) net->op_x_slot[r.first] = tensors[r.second].slot;
}

// ahead of ops [first, last) on `s`: zero the maxima they raise, and take the maxima of the outside tensors they read
static int net_reset_amax(pp_net* net, int first, int last, int batch, hipStream_t s) {
    if (!net->amax) return PP_OK;
    int s0 = -1, s1 = -1;
    // a range that starts (or ends) between the producers of a tracked tensor would keep (or lose) part of its maxima: the scale
    // would then depend on earlier runs, which the fp16 form promises it never does
    for (int k = 0; k < (int)net->slot_first.size(); ++k) {
        const int a = net->slot_first[k], b = net->slot_last[k];
        if (a < 0 || b < first || a >= last) continue;
        PP_REQUIRE(a >= first && b < last, "pp_net_run: op range [%d, %d) splits the producers (ops %d .. %d) of a tensor whose per-sample maxima are tracked",
                   first, last, a, b);
    }
    for (int k = 0; k < (int)net->slot_owner.size(); ++k)
        if (net->slot_owner[k] >= first && net->slot_owner[k] < last) {
            if (s0 < 0) s0 = k;
            s1 = k + 1;
        }
    if (s0 >= 0) PP_HIP_CHECK(hipMemsetAsync(net->amax_slot(s0), 0, (size_t)(s1 - s0) * net->max_batch * sizeof(unsigned), s));
    for (auto& e : net->ext) {
        if (e.last_reader < first || e.first_reader >= last) continue;
        if (e.provided) {        // the caller's producer filled the slot for THIS run (pp_net_input_amax): a one-shot promise
            e.provided = false;
            continue;
        }
        PP_HIP_CHECK(hipMemsetAsync(net->amax_slot(e.slot), 0, (size_t)batch * sizeof(unsigned), s));
        int rc = pp_launch_amax(net->buf_ptr(e.buf), batch, net->buf_elems[e.buf], net->amax_slot(e.slot), s);
        if (rc != PP_OK) return rc;
    }
    return PP_OK;
}

static void net_plan_lanes(pp_net* net, int n_lanes) {
    const int n = (int)net->ops.size(), nb = (int)net->bufs.size();
    net->op_lane.assign(n, 0);
    net->op_waits.assign(n, {});
    net->op_needs_event.assign(n, 0);
    std::vector<int> last_writer(nb, -1);
    std::vector<std::vector<int>> readers(nb);
    std::vector<int> lane_tail(n_lanes, -1);
    for (int i = 0; i < n; ++i) {
        const pp_op& op = net->ops[i];
        std::vector<int> deps;
        auto add = [&](int d) {
            if (d >= 0 && std::find(deps.begin(), deps.end(), d) == deps.end()) deps.push_back(d);
        };
        for (int b : {op.in, op.res1, op.res2, op.in2, op.in3})
            if (b >= 0) add(last_writer[b]);
        add(last_writer[op.out]);                       // WAW
        for (int r : readers[op.out]) add(r);           // WAR
        // lane: continue the lane whose tail is one of the dependencies (largest index), else the stalest lane
        int lane = -1, best = -1;
        for (int l = 0; l < n_lanes; ++l)
            if (lane_tail[l] >= 0 && lane_tail[l] > best && std::find(deps.begin(), deps.end(), lane_tail[l]) != deps.end()) {
                best = lane_tail[l];
                lane = l;
            }
        if (lane < 0) {
            lane = 0;
            for (int l = 1; l < n_lanes; ++l)
                if (lane_tail[l] < lane_tail[lane]) lane = l;
        }
        net->op_lane[i] = lane;
        for (int d : deps)
            if (net->op_lane[d] != lane) {
                net->op_waits[i].push_back(d);
                net->op_needs_event[d] = 1;
            }
        lane_tail[lane] = i;
        for (int b : {op.in, op.res1, op.res2, op.in2, op.in3})
            if (b >= 0 && b != op.out) readers[b].push_back(i);
        last_writer[op.out] = i;
        readers[op.out].clear();
    }
}

static int net_check_op(const pp_net& net, const pp_op& op, int idx) {
    const int nb = (int)net.bufs.size();
    PP_REQUIRE(op.in >= 0 && op.in < nb && op.out >= 0 && op.out < nb, "op %d: buffer id out of range", idx);
    PP_REQUIRE(op.res1 < nb && op.res2 < nb, "op %d: residual buffer id out of range", idx);
    PP_REQUIRE(op.in2 < nb && op.in3 < nb && (op.type == PP_OP_UPSAMPLE_ADD || (op.in2 < 0 && op.in3 < 0)),
               "op %d: in2 / in3 are inputs of PP_OP_UPSAMPLE_ADD only (-1 elsewhere)", idx);
    const pp_buf& bi = net.bufs[op.in];
    const pp_buf& bo = net.bufs[op.out];
    const int eh = op.pad_end & 1, ew = (op.pad_end >> 1) & 1;   // TensorFlow SAME: the odd padding row / column goes last
    PP_REQUIRE(op.out_c_off >= 0 && op.in_c_off >= 0 && (op.out_c_off & 3) == 0 && (op.in_c_off & 3) == 0,
               "op %d: channel offsets must be non-negative multiples of 4", idx);
    if (op.type != PP_OP_CONV) {
        for (int b : {op.in, op.out, op.res1, op.res2, op.in2, op.in3})
            PP_REQUIRE(b < 0 || net.bufs[b].pad == 0, "op %d: only convolutions may touch a buffer with a zero halo (buffer %d)", idx, b);
    } else {
        PP_REQUIRE(bo.pad == 0 || (!op.out_nchw && op.out_c_off == 0 && bo.c == op.cout && (op.cout & 3) == 0),
                   "op %d: a halo output buffer must be a plain NHWC tensor of the op's own channels", idx);
    }
    if (op.type == PP_OP_CONV) {
        PP_REQUIRE(bi.c == op.cin && op.in_c_off == 0, "op %d: in buffer has %d channels, op.cin=%d", idx, bi.c, op.cin);
        PP_REQUIRE(op.out_c_off + op.cout <= bo.c && (op.out_c_off == 0 ? true : !op.out_nchw),
                   "op %d: channels [%d, %d) do not fit the out buffer (%d channels)", idx, op.out_c_off,
                   op.out_c_off + op.cout, bo.c);
        PP_REQUIRE(bo.c == op.cout || ((bo.c & 3) == 0 && !op.out_nchw), "op %d: a sliced out buffer needs c %% 4 == 0", idx);
        PP_REQUIRE(op.relu >= PP_RELU_NONE && op.relu <= PP_ACT_SWISH, "op %d: unknown activation %d", idx, op.relu);
        // bits 2 / 3: padding in front only -- the last output row / column of the symmetric-padding result is not computed
        const int ho = pp_conv_out_dim(bi.h + eh, op.kh, op.stride, op.pad_h, op.dil_h) - ((op.pad_end >> 2) & 1);
        const int wo = pp_conv_out_dim(bi.w + ew, op.kw, op.stride, op.pad_w, op.dil_w) - ((op.pad_end >> 3) & 1);
        PP_REQUIRE((ho << op.up_log2) == bo.h && (wo << op.up_log2) == bo.w,
                   "op %d: conv output %dx%d (<<%d) does not match out buffer %dx%d", idx, ho, wo, op.up_log2,
                   bo.h, bo.w);
        const size_t kpad = ((size_t)op.kh * op.kw * op.cin + 31) / 32 * 32;
        const size_t cpad = ((size_t)op.cout + 15) / 16 * 16;
        PP_REQUIRE(op.w_off >= 0 && (size_t)op.w_off + kpad * cpad <= net.n_weights, "op %d: weights out of blob", idx);
        PP_REQUIRE(op.b_off >= 0 && (size_t)op.b_off + cpad <= net.n_weights, "op %d: bias out of blob", idx);
        PP_REQUIRE((op.w_off % 4) == 0 && (op.b_off % 4) == 0, "op %d: blob offsets must be 16-byte aligned", idx);
        if (op.res1 >= 0) PP_REQUIRE(net.bufs[op.res1].c == op.cout, "op %d: res1 channel mismatch", idx);
        if (op.res2 >= 0)
            PP_REQUIRE(net.bufs[op.res2].c == op.cout && net.bufs[op.res2].h == bo.h && net.bufs[op.res2].w == bo.w,
                       "op %d: res2 shape mismatch", idx);
    } else if (op.type == PP_OP_MAXPOOL) {
        PP_REQUIRE(op.cin == op.cout && (op.cin & 3) == 0 && op.in_c_off + op.cin <= bi.c && op.out_c_off + op.cout <= bo.c &&
                       (bi.c & 3) == 0 && (bo.c & 3) == 0,
                   "op %d: maxpool channel slices do not fit", idx);
        PP_REQUIRE(pp_conv_out_dim(bi.h + eh, op.kh, op.stride, op.pad_h, 1) == bo.h &&
                       pp_conv_out_dim(bi.w + ew, op.kw, op.stride, op.pad_w, 1) == bo.w,
                   "op %d: maxpool output dims mismatch", idx);
    } else if (op.type == PP_OP_AVGPOOL) {
        PP_REQUIRE(op.cin == op.cout && (op.cin & 3) == 0 && bi.c == op.cin && bo.c == op.cin && op.in != op.out && op.kh > 0 &&
                       op.kw > 0 && op.stride > 0 && bi.h >= op.kh && bi.w >= op.kw && (bi.h - op.kh) / op.stride + 1 == bo.h &&
                       (bi.w - op.kw) / op.stride + 1 == bo.w,
                   "op %d: avgpool needs in [h][w][c] and out [(h - kh) / s + 1][(w - kw) / s + 1][c], c %% 4 == 0", idx);
    } else if (op.type == PP_OP_COPY) {
        PP_REQUIRE(bi.c == bo.c && bi.h == bo.h && bi.w == bo.w, "op %d: copy shape mismatch", idx);
    } else if (op.type == PP_OP_DEPTH_TO_SPACE) {
        PP_REQUIRE(op.cout > 0 && (op.cout & 3) == 0 && bi.c == 4 * op.cout && bo.c == op.cout && bo.h == 2 * bi.h && bo.w == 2 * bi.w,
                   "op %d: depth_to_space needs in [h][w][4*cout] and out [2h][2w][cout]", idx);
    } else if (op.type == PP_OP_DECONV_BF16) {
        PP_REQUIRE(bi.c == op.cin && bo.c == op.cout && bo.h == 2 * bi.h && bo.w == 2 * bi.w && op.in != op.out,
                   "op %d: deconv_bf16 needs in [h][w][cin] and out [2h][2w][cout]", idx);
        PP_REQUIRE(op.cin % 64 == 0 && (op.cout & 7) == 0, "op %d: deconv_bf16 needs cin %% 64 == 0 and cout %% 8 == 0", idx);
        PP_REQUIRE(op.relu == PP_RELU_NONE || op.relu == PP_RELU_LAST, "op %d: deconv_bf16 supports PP_RELU_NONE / PP_RELU_LAST", idx);
        PP_REQUIRE(op.w_off >= 0 && (op.w_off % 4) == 0 && (size_t)op.w_off + (size_t)16 * op.cout * op.cin <= net.n_weights &&
                       op.b_off >= 0 && (op.b_off % 4) == 0 && (size_t)op.b_off + op.cout <= net.n_weights,
                   "op %d: deconv_bf16 parameters out of blob", idx);
    } else if (op.type == PP_OP_UPSAMPLE_ADD) {
        PP_REQUIRE(op.cin == op.cout && (op.cout & 3) == 0 && bi.c == op.cout && bo.c == op.cout && op.up_log2 >= 0 &&
                       (bi.h << op.up_log2) == bo.h && (bi.w << op.up_log2) == bo.w && op.in != op.out,
                   "op %d: upsample_add needs in [h][w][c] and out [h << up][w << up][c]", idx);
        PP_REQUIRE(op.relu == PP_RELU_NONE || op.relu == PP_RELU_LAST, "op %d: upsample_add supports PP_RELU_NONE / PP_RELU_LAST", idx);
        for (int r : {op.res1, op.res2})
            if (r >= 0)
                PP_REQUIRE(net.bufs[r].c == bo.c && net.bufs[r].h == bo.h && net.bufs[r].w == bo.w, "op %d: residual shape mismatch", idx);
        PP_REQUIRE(op.in3 < 0 || op.in2 >= 0, "op %d: in3 without in2", idx);
        for (int k = 0; k < 2; ++k) {
            const int b = k ? op.in3 : op.in2, u = k ? op.up3_log2 : op.up2_log2;
            if (b >= 0)
                PP_REQUIRE(u >= 0 && u <= 5 && net.bufs[b].c == bo.c && (net.bufs[b].h << u) == bo.h && (net.bufs[b].w << u) == bo.w &&
                               b != op.out, "op %d: in%d must be [h >> up][w >> up][c] of the out buffer", idx, k + 2);
        }
    } else if (op.type == PP_OP_VIT_ENCODER) {
        PP_REQUIRE(op.cin == op.cout && bi.c == op.cin && bo.c == op.cin && bi.h == bo.h && bi.w == bo.w && op.in != op.out,
                   "op %d: vit encoder needs distinct in / out buffers of [h][w][dim]", idx);
        PP_REQUIRE(op.kh > 0 && op.kw > 0 && op.stride > 0, "op %d: vit encoder needs depth (kh), heads (kw), mlp ratio (stride)", idx);
        PP_REQUIRE(op.w_off >= 0 && (op.w_off % 4) == 0 &&
                       (size_t)op.w_off + pp_vit_param_floats(bi.h * bi.w, op.cin, op.kh, op.cin * op.stride) <= net.n_weights,
                   "op %d: vit encoder parameters out of blob", idx);
    } else {
        pp_set_error("op %d: unsupported op type %d", idx, op.type);
        return PP_ERR_UNSUPPORTED;
    }
    return PP_OK;
}

static ConvArgs net_conv_args(pp_net* net, const pp_op& op, int batch) {
    const pp_buf& bi = net->bufs[op.in];
    const pp_buf& bo = net->bufs[op.out];
    ConvArgs a{};
    a.x = net->buf_ptr(op.in);
    a.y = net->buf_ptr(op.out);
    a.w = net->weights + op.w_off;
    a.bias = net->weights + op.b_off;
    a.res1 = op.res1 >= 0 ? net->buf_ptr(op.res1) : nullptr;
    a.res2 = op.res2 >= 0 ? net->buf_ptr(op.res2) : nullptr;
    a.N = batch; a.Hin = bi.h; a.Win = bi.w; a.Cin = op.cin;
    a.Hout = bo.h >> op.up_log2; a.Wout = bo.w >> op.up_log2;
    a.Cout = op.cout; a.CoutPad = (op.cout + 15) / 16 * 16;
    a.KH = op.kh; a.KW = op.kw; a.stride = op.stride; a.pad_h = op.pad_h; a.pad_w = op.pad_w;
    a.dil_h = op.dil_h; a.dil_w = op.dil_w;
    a.K = op.kh * op.kw * op.cin; a.Kpad = (a.K + 31) / 32 * 32;
    a.HWout = a.Hout * a.Wout; a.M = batch * a.HWout;
    a.relu = op.relu; a.up_log2 = op.up_log2; a.out_nchw = op.out_nchw;
    a.res1_shift = op.res1_shift; a.res1_off_w = op.res1_off_w;
    a.res1_H = op.res1 >= 0 ? net->bufs[op.res1].h : 0;
    a.res1_W = op.res1 >= 0 ? net->bufs[op.res1].w : 0;
    a.y_stride = bo.c; a.y_coff = op.out_c_off;
    a.x_pad = bi.pad; a.y_pad = bo.pad;
    a.r1_pad = op.res1 >= 0 ? net->bufs[op.res1].pad : 0;
    a.r2_pad = op.res2 >= 0 ? net->bufs[op.res2].pad : 0;
    const size_t idx = &op - net->ops.data();
    a.wsplit = (net->wsplit && idx < net->wsplit_off.size() && net->wsplit_off[idx] >= 0) ? net->wsplit + net->wsplit_off[idx] : nullptr;
    a.numerics = net->numerics;
    a.split_f16 = net->split_f16;
    if (idx < net->op_x_slot.size()) {
        a.x_amax = net->amax_slot(net->op_x_slot[idx]);
        a.y_amax = net->amax_slot(net->op_y_slot[idx]);
    }
    return a;
}

static int net_launch_op_body(pp_net* net, const pp_op& op, int batch, hipStream_t s);

static int net_launch_op(pp_net* net, const pp_op& op, int batch, hipStream_t s) {
    const size_t idx = &op - net->ops.data();
    int rc = net_launch_op_body(net, op, batch, s);
    if (rc == PP_OK && net->amax && net->op_post_slot[idx] >= 0)
        rc = pp_launch_amax(net->buf_ptr(op.out), batch, net->buf_elems[op.out], net->amax_slot(net->op_post_slot[idx]), s);
    return rc;
}

static int net_launch_op_body(pp_net* net, const pp_op& op, int batch, hipStream_t s) {
    const pp_buf& bi = net->bufs[op.in];
    const pp_buf& bo = net->bufs[op.out];
    if (op.type == PP_OP_CONV) {
        const ConvArgs a = net_conv_args(net, op, batch);
        return pp_launch_conv(a, s);
    } else if (op.type == PP_OP_MAXPOOL) {
        PoolArgs p{};
        p.x = net->buf_ptr(op.in); p.y = net->buf_ptr(op.out);
        p.N = batch; p.Hin = bi.h; p.Win = bi.w; p.C = op.cin; p.Hout = bo.h; p.Wout = bo.w;
        p.x_stride = bi.c; p.x_coff = op.in_c_off; p.y_stride = bo.c; p.y_coff = op.out_c_off;
        p.KH = op.kh; p.KW = op.kw; p.stride = op.stride; p.pad_h = op.pad_h; p.pad_w = op.pad_w;
        return pp_launch_maxpool(p, s);
    } else if (op.type == PP_OP_AVGPOOL) {
        return pp_launch_avgpool(net->buf_ptr(op.in), net->buf_ptr(op.out), batch, bi.h, bi.w, op.cin, op.kh, op.kw, op.stride, s);
    } else if (op.type == PP_OP_COPY) {
        PP_HIP_CHECK(hipMemcpyAsync(net->buf_ptr(op.out), net->buf_ptr(op.in),
                                    (size_t)batch * net->buf_elems[op.in] * sizeof(float),
                                    hipMemcpyDeviceToDevice, s));
        return PP_OK;
    } else if (op.type == PP_OP_DEPTH_TO_SPACE) {
        return pp_launch_depth_to_space(net->buf_ptr(op.in), net->buf_ptr(op.out), batch, bi.h, bi.w, op.cout, s);
    } else if (op.type == PP_OP_DECONV_BF16) {
        return pp_deconv_bf16_run(net->deconvs[&op - net->ops.data()], net->buf_ptr(op.in), net->buf_ptr(op.out), batch,
                                  op.relu == PP_RELU_LAST, s);
    } else if (op.type == PP_OP_UPSAMPLE_ADD) {
        return pp_launch_upsample_add(net->buf_ptr(op.in), op.res1 >= 0 ? net->buf_ptr(op.res1) : nullptr,
                                      op.res2 >= 0 ? net->buf_ptr(op.res2) : nullptr, net->buf_ptr(op.out), batch, bo.h, bo.w,
                                      bo.c, op.up_log2, op.relu == PP_RELU_LAST, s, op.in2 >= 0 ? net->buf_ptr(op.in2) : nullptr,
                                      op.up2_log2, op.in3 >= 0 ? net->buf_ptr(op.in3) : nullptr, op.up3_log2,
                                      net->amax && net->op_y_slot[&op - net->ops.data()] >= 0 ? net->amax_slot(net->op_y_slot[&op - net->ops.data()]) : nullptr);
    } else if (op.type == PP_OP_VIT_ENCODER) {
        pp_vit_encoder* enc = net->vits[&op - net->ops.data()];
        return pp_vit_encoder_run(enc, net->buf_ptr(op.in), net->buf_ptr(op.out), batch, s);
    }
    return PP_ERR_UNSUPPORTED;
}

int pp_net_dims(pp_net* net, int buf, int* h, int* w, int* c) {
    if (!net || buf < 0 || buf >= (int)net->bufs.size()) return PP_ERR_ARG;
    *h = net->bufs[buf].h; *w = net->bufs[buf].w; *c = net->bufs[buf].c;
    return PP_OK;
}
int pp_net_max_batch(pp_net* net) { return net ? net->max_batch : 0; }
pp_ctx* pp_net_ctx(pp_net* net) { return net ? net->ctx : nullptr; }

unsigned* pp_net_input_amax_slot(pp_net* net, int buf) {
    if (!net || !net->amax) return nullptr;
    for (auto& e : net->ext)
        if (e.buf == buf) return net->amax_slot(e.slot);
    return nullptr;
}

extern "C" {

int pp_net_create(pp_ctx* ctx, const pp_op* ops, int n_ops, const pp_buf* bufs, int n_bufs,
                  const float* weights, size_t n_weights, int max_batch, pp_net** out) {
    return pp_net_create_mem(ctx, ops, n_ops, bufs, n_bufs, weights, n_weights, PP_MEM_HOST, max_batch, out);
}

int pp_net_create_mem(pp_ctx* ctx, const pp_op* ops, int n_ops, const pp_buf* bufs, int n_bufs,
                      const float* weights, size_t n_weights, int weights_mem, int max_batch, pp_net** out) {
    return pp_net_create_ex(ctx, ops, n_ops, bufs, n_bufs, weights, n_weights, weights_mem, max_batch, PP_NET_NUMERICS_DEFAULT, out);
}

int pp_net_numerics(pp_net* net) { return net ? net->numerics : PP_ERR_ARG; }
int pp_net_split_kind(pp_net* net) {
    if (!net) return PP_ERR_ARG;
    return net->numerics != PP_NET_NUMERICS_SPLIT ? PP_NET_NUMERICS_EXACT : net->split_f16 ? PP_NET_NUMERICS_SPLIT_F16 : PP_NET_NUMERICS_SPLIT_BF16;
}

int pp_net_create_ex(pp_ctx* ctx, const pp_op* ops, int n_ops, const pp_buf* bufs, int n_bufs,
                     const float* weights, size_t n_weights, int weights_mem, int max_batch, int numerics, pp_net** out) {
    PP_REQUIRE(ctx && ops && bufs && weights && out, "pp_net_create: NULL argument");
    PP_REQUIRE(numerics >= PP_NET_NUMERICS_DEFAULT && numerics <= PP_NET_NUMERICS_SPLIT_F16, "pp_net_create_ex: bad numerics %d", numerics);
    PP_REQUIRE(weights_mem == PP_MEM_HOST || weights_mem == PP_MEM_DEVICE, "pp_net_create: bad weights_mem %d", weights_mem);
    PP_REQUIRE(n_ops > 0 && n_bufs > 0 && max_batch > 0, "pp_net_create: empty program");
    *out = nullptr;
    std::unique_ptr<pp_net> net(new pp_net());
    net->ctx = ctx;
    net->ops.assign(ops, ops + n_ops);
    net->bufs.assign(bufs, bufs + n_bufs);
    net->max_batch = max_batch;
    net->n_weights = n_weights;
    // the process-wide switch is read HERE, once: later pp_conv_exact calls (or other threads) do not change this net
    net->numerics = numerics != PP_NET_NUMERICS_DEFAULT ? numerics : (pp_conv_split_enabled() ? PP_NET_NUMERICS_SPLIT : PP_NET_NUMERICS_EXACT);
    if (net->numerics >= PP_NET_NUMERICS_SPLIT) {       // the split form: named, or the process-wide default at this moment
        net->split_f16 = net->numerics == PP_NET_NUMERICS_SPLIT ? pp_conv_split_f16_default() : net->numerics == PP_NET_NUMERICS_SPLIT_F16;
        net->numerics = PP_NET_NUMERICS_SPLIT;
    }
    size_t off = 0;
    for (int b = 0; b < n_bufs; ++b) {
        PP_REQUIRE(bufs[b].h > 0 && bufs[b].w > 0 && bufs[b].c > 0 && bufs[b].pad >= 0 && bufs[b].pad <= 8, "buffer %d has an empty dim / bad halo", b);
        const size_t e = (size_t)(bufs[b].h + bufs[b].pad) * (bufs[b].w + bufs[b].pad) * bufs[b].c;   // halo: never written, stays zero
        net->buf_elems.push_back(e);
        net->buf_off.push_back(off);
        off += (e * max_batch + 63) / 64 * 64;   // 256-byte aligned
    }
    net->arena_floats = off;
    for (int i = 0; i < n_ops; ++i) {
        int rc = net_check_op(*net, net->ops[i], i);
        if (rc != PP_OK) return rc;
    }
    PP_HIP_CHECK(hipSetDevice(ctx->device));
    PP_HIP_CHECK(hipMalloc((void**)&net->weights, n_weights * sizeof(float)));
    PP_HIP_CHECK(hipMemcpyAsync(net->weights, weights, n_weights * sizeof(float),
                                weights_mem == PP_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, ctx->stream));
    PP_HIP_CHECK(hipMalloc((void**)&net->arena, net->arena_floats * sizeof(float)));
    PP_HIP_CHECK(hipMemsetAsync(net->arena, 0, net->arena_floats * sizeof(float), ctx->stream));
    net->vits.assign(n_ops, nullptr);
    net->deconvs.assign(n_ops, nullptr);
    for (int i = 0; i < n_ops; ++i) {
        const pp_op& op = net->ops[i];
        if (op.type == PP_OP_DECONV_BF16) {
            const pp_buf& bi = net->bufs[op.in];
            int rc = pp_deconv_bf16_create(net->weights + op.w_off, net->weights + op.b_off, bi.h, bi.w, op.cin, op.cout,
                                           max_batch, ctx->stream, &net->deconvs[i]);
            if (rc != PP_OK) {
                pp_net_destroy(net.release());
                return rc;
            }
        }
        if (op.type != PP_OP_VIT_ENCODER) continue;
        const pp_buf& bi = net->bufs[op.in];
        int rc = pp_vit_encoder_create(net->weights + op.w_off, bi.h * bi.w, op.cin, op.kh, op.kw, op.cin * op.stride,
                                       max_batch, ctx->stream, &net->vits[i]);
        if (rc != PP_OK) {
            pp_net_destroy(net.release());
            return rc;
        }
    }
    // bf16-split copies of the weights of every conv the split kernel will run
    net->wsplit_off.assign(n_ops, -1);
    if (net->numerics == PP_NET_NUMERICS_SPLIT) {
        size_t bytes = 0;
        for (int i = 0; i < n_ops; ++i) {
            if (net->ops[i].type != PP_OP_CONV) continue;
            const ConvArgs a = net_conv_args(net.get(), net->ops[i], 1);
            if (!pp_conv_split_eligible(a)) continue;
            net->wsplit_off[i] = (long long)bytes;
            bytes += (pp_conv_split_bytes(a) + 255) / 256 * 256;
        }
        net_plan_amax(net.get());
        if (!net->slot_owner.empty()) {
            const size_t ab = net->slot_owner.size() * (size_t)max_batch * sizeof(unsigned);
            PP_HIP_CHECK(hipMalloc((void**)&net->amax, ab));
            PP_HIP_CHECK(hipMemsetAsync(net->amax, 0, ab, ctx->stream));
        }
        if (bytes) {
            PP_HIP_CHECK(hipMalloc((void**)&net->wsplit, bytes));
            for (int i = 0; i < n_ops; ++i) {
                if (net->wsplit_off[i] < 0) continue;
                const ConvArgs a = net_conv_args(net.get(), net->ops[i], 1);
                int rc = pp_conv_split_weights(a, net->wsplit + net->wsplit_off[i], ctx->stream);
                if (rc != PP_OK) return rc;
            }
        }
    }
    PP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    // per-geometry tables of the pipelined conv kernel: built here, never inside a launch that may be under graph capture
    for (int i = 0; i < n_ops; ++i)
        if (net->ops[i].type == PP_OP_CONV) {
            int rc = pp_conv_prepare(net_conv_args(net.get(), net->ops[i], 1));
            if (rc != PP_OK) return rc;
        }
    const char* env_lanes = getenv("POSEPIPE_NET_LANES");
    const int n_lanes = env_lanes ? atoi(env_lanes) : 4;
    int rc_l = net_make_lanes(net.get(), n_lanes);
    if (rc_l != PP_OK) return rc_l;
    *out = net.release();
    return PP_OK;
}

// (re)build the lane streams, the op -> lane plan and its events for `n_lanes` (<= 1, or a program under 8 ops: none -- every op on the
// ctx stream).  The caller has synchronised: nothing of the net is in flight.
static int net_make_lanes(pp_net* net, int n_lanes) {
    for (auto l : net->lanes) (void)hipStreamDestroy(l);
    net->lanes.clear();
    for (auto e : net->op_done)
        if (e) (void)hipEventDestroy(e);
    net->op_done.clear();
    for (auto e : net->join_ev)
        if (e) (void)hipEventDestroy(e);
    net->join_ev.clear();
    if (net->fork_ev) (void)hipEventDestroy(net->fork_ev);
    net->fork_ev = nullptr;
    const int n_ops = (int)net->ops.size();
    if (n_lanes > 1 && n_ops >= 8) {
        net_plan_lanes(net, n_lanes);
        net->lanes.resize(n_lanes);
        for (auto& l : net->lanes) PP_HIP_CHECK(hipStreamCreateWithFlags(&l, hipStreamNonBlocking));
        net->op_done.assign(n_ops, nullptr);
        for (int i = 0; i < n_ops; ++i)
            if (net->op_needs_event[i]) PP_HIP_CHECK(hipEventCreateWithFlags(&net->op_done[i], hipEventDisableTiming));
        PP_HIP_CHECK(hipEventCreateWithFlags(&net->fork_ev, hipEventDisableTiming));
        net->join_ev.resize(n_lanes);
        for (auto& e : net->join_ev) PP_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    return PP_OK;
}

int pp_net_input_amax(pp_net* net, int buf, void** dptr) {
    PP_REQUIRE(net && dptr && buf >= 0 && buf < (int)net->bufs.size(), "pp_net_input_amax: bad argument");
    *dptr = nullptr;
    for (auto& e : net->ext)
        if (e.buf == buf && net->amax) {
            e.provided = true;
            *dptr = net->amax_slot(e.slot);
        }
    return PP_OK;
}

int pp_net_conv_kinds(pp_net* net, int* kinds) {
    PP_REQUIRE(net && kinds, "pp_net_conv_kinds: NULL argument");
    for (size_t i = 0; i < net->ops.size(); ++i) {
        if (net->ops[i].type != PP_OP_CONV) {
            kinds[i] = 0;
            continue;
        }
        kinds[i] = (net->numerics == PP_NET_NUMERICS_SPLIT && net->wsplit && net->wsplit_off[i] >= 0) ? 2 : 1;
    }
    return PP_OK;
}

void pp_net_destroy(pp_net* net) {
    if (!net) return;
    if (net->ctx && net->ctx->stream) (void)hipStreamSynchronize(net->ctx->stream);
    for (auto l : net->lanes) {
        (void)hipStreamSynchronize(l);
        (void)hipStreamDestroy(l);
    }
    for (auto e : net->op_done)
        if (e) (void)hipEventDestroy(e);
    for (auto e : net->join_ev)
        if (e) (void)hipEventDestroy(e);
    if (net->fork_ev) (void)hipEventDestroy(net->fork_ev);
    if (net->graph_exec) (void)hipGraphExecDestroy(net->graph_exec);
    for (auto* v : net->vits) pp_vit_encoder_destroy(v);
    for (auto* d : net->deconvs) pp_deconv_bf16_destroy(d);
    if (net->weights) (void)hipFree(net->weights);
    if (net->wsplit) (void)hipFree(net->wsplit);
    if (net->amax) (void)hipFree(net->amax);
    if (net->arena) (void)hipFree(net->arena);
    delete net;
}

int pp_net_buffer(pp_net* net, int buf, void** dptr, size_t* bytes_per_sample) {
    PP_REQUIRE(net && buf >= 0 && buf < (int)net->bufs.size(), "pp_net_buffer: bad buffer id");
    PP_REQUIRE(net->bufs[buf].pad == 0, "pp_net_buffer: buffer %d has a zero halo (conv-internal layout); expose dense buffers only", buf);
    if (dptr) *dptr = net->buf_ptr(buf);
    if (bytes_per_sample) *bytes_per_sample = net->buf_elems[buf] * sizeof(float);
    return PP_OK;
}

int pp_net_run(pp_net* net, int batch, int first_op, int last_op) {
    PP_REQUIRE(net, "pp_net_run: net is NULL");
    PP_REQUIRE(batch > 0 && batch <= net->max_batch, "pp_net_run: batch %d not in (0,%d]", batch, net->max_batch);
    if (last_op < 0) last_op = (int)net->ops.size();
    PP_REQUIRE(first_op >= 0 && last_op <= (int)net->ops.size() && first_op <= last_op, "pp_net_run: bad op range");
    if (net->graph_exec && batch == net->graph_batch && first_op == 0 && last_op == (int)net->ops.size()) {
        PP_HIP_CHECK(hipGraphLaunch(net->graph_exec, net->ctx->stream));
        return PP_OK;
    }
    hipStream_t main = net->ctx->stream;
    PpRange range("pp_net_run");
    {
        int rc = net_reset_amax(net, first_op, last_op, batch, main);
        if (rc != PP_OK) return rc;
    }
    if (net->lanes.empty() || !net->use_lanes || last_op - first_op < 8) {
        for (int i = first_op; i < last_op; ++i) {
            int rc = net_launch_op(net, net->ops[i], batch, main);
            if (rc != PP_OK) return rc;
        }
        return PP_OK;
    }
    // fork: every lane starts after whatever is already queued on the ctx stream
    PP_HIP_CHECK(hipEventRecord(net->fork_ev, main));
    for (hipStream_t l : net->lanes) PP_HIP_CHECK(hipStreamWaitEvent(l, net->fork_ev, 0));
    for (int i = first_op; i < last_op; ++i) {
        hipStream_t s = net->lanes[net->op_lane[i]];
        for (int d : net->op_waits[i])
            if (d >= first_op) PP_HIP_CHECK(hipStreamWaitEvent(s, net->op_done[d], 0));
        int rc = net_launch_op(net, net->ops[i], batch, s);
        if (rc != PP_OK) return rc;
        if (net->op_needs_event[i]) PP_HIP_CHECK(hipEventRecord(net->op_done[i], s));
    }
    // join: the ctx stream continues after every lane
    for (size_t l = 0; l < net->lanes.size(); ++l) {
        PP_HIP_CHECK(hipEventRecord(net->join_ev[l], net->lanes[l]));
        PP_HIP_CHECK(hipStreamWaitEvent(main, net->join_ev[l], 0));
    }
    return PP_OK;
}

int pp_net_vit_timing(pp_net* net, int enable, float* ms3, int* n_gemm) {
    PP_REQUIRE(net, "pp_net_vit_timing: net is NULL");
    for (auto* v : net->vits) {
        if (!v) continue;
        if (ms3) {
            PP_HIP_CHECK(hipStreamSynchronize(net->ctx->stream));
            int rc = pp_vit_encoder_get_timing(v, ms3, n_gemm);
            if (rc != PP_OK) return rc;
        }
        pp_vit_encoder_set_timing(v, enable);
        return PP_OK;
    }
    pp_set_error("pp_net_vit_timing: the program has no PP_OP_VIT_ENCODER");
    return PP_ERR_STATE;
}

int pp_net_set_lane_count(pp_net* net, int n_lanes) {
    PP_REQUIRE(net && n_lanes >= 1 && n_lanes <= 16, "pp_net_set_lane_count: net is NULL or n_lanes not in [1, 16]");
    PP_HIP_CHECK(hipStreamSynchronize(net->ctx->stream));
    for (auto l : net->lanes) PP_HIP_CHECK(hipStreamSynchronize(l));
    if (net->graph_exec) {            // a captured graph holds the old plan
        PP_HIP_CHECK(hipGraphExecDestroy(net->graph_exec));
        net->graph_exec = nullptr;
    }
    return net_make_lanes(net, n_lanes);
}

int pp_net_set_lanes(pp_net* net, int enable) {
    PP_REQUIRE(net, "pp_net_set_lanes: net is NULL");
    PP_HIP_CHECK(hipStreamSynchronize(net->ctx->stream));
    net->use_lanes = enable != 0;
    return PP_OK;
}

int pp_net_capture(pp_net* net, int batch) {
    PP_REQUIRE(net, "pp_net_capture: net is NULL");
    PP_REQUIRE(batch > 0 && batch <= net->max_batch, "pp_net_capture: bad batch");
    if (net->graph_exec) {
        PP_HIP_CHECK(hipGraphExecDestroy(net->graph_exec));
        net->graph_exec = nullptr;
    }
    hipStream_t s = net->ctx->stream;
    PP_HIP_CHECK(hipStreamSynchronize(s));
    hipGraph_t graph = nullptr;
    PP_HIP_CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    int rc = net_reset_amax(net, 0, (int)net->ops.size(), batch, s);
    for (size_t i = 0; i < net->ops.size() && rc == PP_OK; ++i) rc = net_launch_op(net, net->ops[i], batch, s);
    hipError_t e = hipStreamEndCapture(s, &graph);
    if (rc != PP_OK) {
        if (graph) (void)hipGraphDestroy(graph);
        return rc;
    }
    PP_HIP_CHECK(e);
    PP_HIP_CHECK(hipGraphInstantiate(&net->graph_exec, graph, nullptr, nullptr, 0));
    PP_HIP_CHECK(hipGraphDestroy(graph));
    net->graph_batch = batch;
    return PP_OK;
}

int pp_net_forward(pp_net* net, int batch, int in_buf, const float* in, int out_buf, float* out, int mem) {
    PP_REQUIRE(net && in && out, "pp_net_forward: NULL argument");
    PP_REQUIRE(in_buf >= 0 && in_buf < (int)net->bufs.size() && out_buf >= 0 && out_buf < (int)net->bufs.size(),
               "pp_net_forward: bad buffer id");
    PP_REQUIRE(batch > 0 && batch <= net->max_batch, "pp_net_forward: batch %d not in (0,%d]", batch, net->max_batch);
    PP_REQUIRE(net->bufs[in_buf].pad == 0 && net->bufs[out_buf].pad == 0, "pp_net_forward: input / output buffers must be dense (pad 0)");
    hipStream_t s = net->ctx->stream;
    const hipMemcpyKind kin = mem == PP_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
    const hipMemcpyKind kout = mem == PP_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    PP_HIP_CHECK(hipMemcpyAsync(net->buf_ptr(in_buf), in, (size_t)batch * net->buf_elems[in_buf] * sizeof(float), kin, s));
    for (auto& e : net->ext)      // the input was just overwritten here: a pending promise of maxima (pp_net_input_amax) is void
        if (e.buf == in_buf) e.provided = false;
    int rc = pp_net_run(net, batch, 0, (int)net->ops.size());
    if (rc != PP_OK) return rc;
    PP_HIP_CHECK(hipMemcpyAsync(out, net->buf_ptr(out_buf), (size_t)batch * net->buf_elems[out_buf] * sizeof(float), kout, s));
    PP_HIP_CHECK(hipStreamSynchronize(s));
    return PP_OK;
}

int pp_net_profile(pp_net* net, int batch, float* ms_per_op) {
    PP_REQUIRE(net, "pp_net_profile: net is NULL");
    PP_REQUIRE(batch > 0 && batch <= net->max_batch, "pp_net_profile: bad batch");
    hipStream_t s = net->ctx->stream;
    const size_t n = net->ops.size();
    std::vector<hipEvent_t> ev(n + 1);
    for (auto& e : ev) PP_HIP_CHECK(hipEventCreate(&e));
    int rc = net_reset_amax(net, 0, (int)n, batch, s);
    PP_HIP_CHECK(hipEventRecord(ev[0], s));
    for (size_t i = 0; i < n && rc == PP_OK; ++i) {
        rc = net_launch_op(net, net->ops[i], batch, s);
        if (rc == PP_OK && hipEventRecord(ev[i + 1], s) != hipSuccess) rc = PP_ERR_HIP;
    }
    if (rc == PP_OK && hipStreamSynchronize(s) != hipSuccess) rc = PP_ERR_HIP;
    if (rc == PP_OK && ms_per_op) {
        for (size_t i = 0; i < n; ++i) {
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, ev[i], ev[i + 1]);
            ms_per_op[i] = ms;
        }
    }
    for (auto& e : ev) (void)hipEventDestroy(e);
    return rc;
}

int pp_conv2d(pp_ctx* ctx, const pp_op* op, int n, int hin, int win, const float* x, const float* w,
              const float* bias, const float* res1, const float* res2, float* y, int res1_h, int res1_w,
              int mem) {
    PP_REQUIRE(ctx && op && x && w && bias && y, "pp_conv2d: NULL argument");
    PP_REQUIRE(op->type == PP_OP_CONV, "pp_conv2d: op is not a conv");
    PP_REQUIRE(n > 0 && hin > 0 && win > 0, "pp_conv2d: empty input");
    ConvArgs a{};
    a.N = n; a.Hin = hin; a.Win = win; a.Cin = op->cin;
    a.Hout = pp_conv_out_dim(hin + (op->pad_end & 1), op->kh, op->stride, op->pad_h, op->dil_h) - ((op->pad_end >> 2) & 1);
    a.Wout = pp_conv_out_dim(win + ((op->pad_end >> 1) & 1), op->kw, op->stride, op->pad_w, op->dil_w) - ((op->pad_end >> 3) & 1);
    PP_REQUIRE(a.Hout > 0 && a.Wout > 0, "pp_conv2d: empty output");
    a.Cout = op->cout; a.CoutPad = (op->cout + 15) / 16 * 16;
    a.KH = op->kh; a.KW = op->kw; a.stride = op->stride; a.pad_h = op->pad_h; a.pad_w = op->pad_w;
    a.dil_h = op->dil_h; a.dil_w = op->dil_w;
    a.K = op->kh * op->kw * op->cin; a.Kpad = (a.K + 31) / 32 * 32;
    a.HWout = a.Hout * a.Wout; a.M = n * a.HWout;
    a.relu = op->relu; a.up_log2 = op->up_log2; a.out_nchw = op->out_nchw;
    a.res1_shift = op->res1_shift; a.res1_off_w = op->res1_off_w;
    const int Ho2 = a.Hout << op->up_log2, Wo2 = a.Wout << op->up_log2;
    a.res1_H = res1 ? (res1_h > 0 ? res1_h : Ho2) : 0;
    a.res1_W = res1 ? (res1_w > 0 ? res1_w : Wo2) : 0;
    a.y_stride = op->cout; a.y_coff = 0;
    PP_REQUIRE(op->out_c_off == 0 && op->in_c_off == 0, "pp_conv2d: channel slices need a layer program");
    const size_t x_e = (size_t)n * hin * win * op->cin;
    const size_t w_e = (size_t)a.Kpad * a.CoutPad;
    const size_t b_e = a.CoutPad;
    const size_t y_e = (size_t)n * Ho2 * Wo2 * op->cout;
    const size_t r1_e = res1 ? (size_t)n * a.res1_H * a.res1_W * op->cout : 0;
    const size_t r2_e = res2 ? y_e : 0;
    if (mem == PP_MEM_DEVICE) {
        a.x = x; a.w = w; a.bias = bias; a.res1 = res1; a.res2 = res2; a.y = y;
        return pp_launch_conv(a, ctx->stream);
    }
    size_t total = 0;
    for (size_t e : {x_e, w_e, b_e, y_e, r1_e, r2_e}) total += ScratchCursor::align(e * sizeof(float));
    int rc = ctx->ensure_scratch(total);
    if (rc != PP_OK) return rc;
    ScratchCursor cur(ctx);
    float* dx = cur.take<float>(x_e);
    float* dw = cur.take<float>(w_e);
    float* db = cur.take<float>(b_e);
    float* dy = cur.take<float>(y_e);
    float* dr1 = cur.take<float>(r1_e);
    float* dr2 = cur.take<float>(r2_e);
    hipStream_t s = ctx->stream;
    PP_HIP_CHECK(hipMemcpyAsync(dx, x, x_e * 4, hipMemcpyHostToDevice, s));
    PP_HIP_CHECK(hipMemcpyAsync(dw, w, w_e * 4, hipMemcpyHostToDevice, s));
    PP_HIP_CHECK(hipMemcpyAsync(db, bias, b_e * 4, hipMemcpyHostToDevice, s));
    if (res1) PP_HIP_CHECK(hipMemcpyAsync(dr1, res1, r1_e * 4, hipMemcpyHostToDevice, s));
    if (res2) PP_HIP_CHECK(hipMemcpyAsync(dr2, res2, r2_e * 4, hipMemcpyHostToDevice, s));
    a.x = dx; a.w = dw; a.bias = db; a.res1 = res1 ? dr1 : nullptr; a.res2 = res2 ? dr2 : nullptr; a.y = dy;
    rc = pp_launch_conv(a, s);
    if (rc != PP_OK) return rc;
    PP_HIP_CHECK(hipMemcpyAsync(y, dy, y_e * 4, hipMemcpyDeviceToHost, s));
    PP_HIP_CHECK(hipStreamSynchronize(s));
    return PP_OK;
}

}  // extern "C"
