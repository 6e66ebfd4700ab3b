// Launch programs: the kernel launches of one model step recorded once, re-issued by ONE host call.
//
// Why.  At batch 1 the hot path is a chain of ~50 short kernels per keyframe (volume plan + volume + lowest cost + mask +
// CVEncoder + decoder + heads).  Enqueued from the host language one entry point at a time, the host spends 0.8 ms per
// keyframe on planning, argument marshalling and tensor bookkeeping around launches that take 1.3 ms on the GPU; a hipGraph
// replay of the same chain measured SLOWER on the wall than the eager launches on this runtime.  A
// program keeps what a graph keeps -- kernel address, geometry, argument bytes -- and replays it as plain hipLaunchKernel
// calls on the caller's stream: the host cost of a step is the launches themselves.
//
// How.  dt_program_begin(stream) puts the calling thread in recording mode for that stream.  Every launch the library then
// makes on it (dt::launch in common.hpp, under every entry point) still executes and is appended to the program; launches on
// other streams are not touched.  dt_program_input registers the device ranges that will differ between replays (the
// step's input tensors): at dt_program_end every POINTER of the recorded arguments (pointer parameters, and the pointer members
// that by-value argument structs declare with DT_ARG_POINTERS: common.hpp) that points into such a range becomes a
// patch (slot, offset), and dt_program_launch rewrites those words from the pointers it is given.  Everything else the
// recorded launches point at -- intermediates, outputs, packed weights, the library's per-stream scratch -- must stay
// allocated for the life of the program: the caller records into buffers it keeps (utils/program.py: a private memory pool).
// dt_program_mark cuts the program into segments that can be launched separately (the caller enqueues an event, or work on
// another stream, between them).  A program replays on the stream it was recorded on (the library's split-K scratch is per
// stream); one replay at a time per program.
#include <cstring>
#include <mutex>
#include <set>
#include <vector>

#include "common.hpp"

namespace dt {

struct ProgNode {
  const void* func;
  dim3 grid, block;
  unsigned shmem;
  unsigned arg_begin;   // first entry of Program::arg_off
  unsigned nargs;
};

struct ProgPatch {
  unsigned blob_off;  // 8-byte aligned offset of the pointer word inside Program::blob
  int slot;
  int64_t delta;      // recorded pointer - recorded base of the slot
};

struct ProgInput {
  uintptr_t base;
  int64_t bytes;
};

struct Program {
  hipStream_t stream = nullptr;
  int device = 0;
  std::vector<ProgNode> nodes;
  std::vector<unsigned> arg_off;         // blob offset of every argument of every node
  std::vector<unsigned> ptr_off;         // blob offsets of the words that ARE device pointers (declared per argument type)
  std::vector<char> blob;                // argument bytes (16-byte aligned base: std::vector<char> of an over-aligned chunk)
  std::vector<void*> arg_ptrs;           // built at end(): &blob[arg_off[i]]
  std::vector<ProgInput> inputs;
  std::vector<ProgPatch> patches;
  std::vector<unsigned> seg_begin;       // node index at which segment i starts (seg_begin[0] = 0)
  long lookalikes = 0;                   // see dt_program_end
};

static thread_local Program* t_rec = nullptr;

static std::mutex g_prog_mutex;
static std::set<Program*> g_programs;  // live handles (dt_program_launch / _free validate against it)

bool recording_on(hipStream_t s) { return t_rec != nullptr && t_rec->stream == s; }

void record_node(const void* func, dim3 grid, dim3 block, size_t shmem, int nargs, const void* const* arg_ptrs,
                 const size_t* arg_sizes, const size_t* arg_aligns, const int* arg_nptrs, const size_t* ptr_offsets) {
  Program* p = t_rec;
  ProgNode n;
  n.func = func;
  n.grid = grid;
  n.block = block;
  n.shmem = (unsigned)shmem;
  n.arg_begin = (unsigned)p->arg_off.size();
  n.nargs = (unsigned)nargs;
  for (int i = 0; i < nargs; ++i) {
    const size_t al = arg_aligns[i] > 8 ? arg_aligns[i] : 8;
    size_t off = (p->blob.size() + al - 1) / al * al;
    const size_t padded = (arg_sizes[i] + 7) / 8 * 8;
    p->blob.resize(off + padded, 0);
    memcpy(p->blob.data() + off, arg_ptrs[i], arg_sizes[i]);
    p->arg_off.push_back((unsigned)off);
    // the pointer members of this argument, by position (common.hpp: arg_pointers) -- the only words a patch may touch
    for (int k = 0; k < arg_nptrs[i]; ++k) p->ptr_off.push_back((unsigned)(off + ptr_offsets[(size_t)i * kMaxArgPointers + k]));
  }
  p->nodes.push_back(n);
}

static Program* live(dt_program_t h) {
  Program* p = reinterpret_cast<Program*>(h);
  std::lock_guard<std::mutex> lock(g_prog_mutex);
  return g_programs.count(p) ? p : nullptr;
}

}  // namespace dt

using namespace dt;

extern "C" {

int dt_program_begin(dt_stream_t s) {
  DT_REQUIRE(t_rec == nullptr, "dt_program_begin: this thread is already recording a program");
  Program* p = new Program();
  p->stream = to_stream(s);
  if (hipGetDevice(&p->device) != hipSuccess) {
    (void)hipGetLastError();
    delete p;
    return fail("dt_program_begin: no current device");
  }
  // blob offsets are relative to an aligned origin: reserve so that early growth does not reallocate too often
  p->blob.reserve(1 << 16);
  p->seg_begin.push_back(0);
  t_rec = p;
  return 0;
}

int dt_program_input(const void* base, int64_t bytes) {
  if (t_rec == nullptr) {
    fail("dt_program_input: no recording in progress on this thread");
    return -1;
  }
  if (base == nullptr || bytes <= 0) {
    fail("dt_program_input: empty range");
    return -1;
  }
  const uintptr_t b = reinterpret_cast<uintptr_t>(base);
  for (const ProgInput& in : t_rec->inputs)
    if (b < in.base + (uintptr_t)in.bytes && in.base < b + (uintptr_t)bytes) {
      fail("dt_program_input: range overlaps input slot already registered (inputs of a program must not alias)");
      return -1;
    }
  t_rec->inputs.push_back(ProgInput{b, bytes});
  return (int)t_rec->inputs.size() - 1;
}

int dt_program_mark(void) {
  if (t_rec == nullptr) {
    fail("dt_program_mark: no recording in progress on this thread");
    return -1;
  }
  t_rec->seg_begin.push_back((unsigned)t_rec->nodes.size());
  return (int)t_rec->seg_begin.size() - 1;
}

int dt_program_abort(void) {
  delete t_rec;
  t_rec = nullptr;
  return 0;
}

int dt_program_end(dt_program_t* out) {
  DT_REQUIRE(t_rec != nullptr, "dt_program_end: no recording in progress on this thread");
  DT_REQUIRE(out != nullptr, "dt_program_end: null output");
  Program* p = t_rec;
  t_rec = nullptr;
  // The blob must start on a 16-byte boundary for by-value arguments with 16-byte alignment (vector types): std::vector<char>
  // of this size comes from operator new (16-byte aligned on this ABI); checked rather than assumed.
  if ((reinterpret_cast<uintptr_t>(p->blob.data()) & 15u) != 0 && !p->blob.empty()) {
    delete p;
    return fail("dt_program_end: argument storage is not 16-byte aligned");
  }
  p->arg_ptrs.resize(p->arg_off.size());
  for (size_t i = 0; i < p->arg_off.size(); ++i) p->arg_ptrs[i] = p->blob.data() + p->arg_off[i];
  // patch table: every declared pointer word that points into a registered input range
  if (!p->inputs.empty()) {
    for (const unsigned off : p->ptr_off) {
      uint64_t v;
      memcpy(&v, p->blob.data() + off, 8);
      for (size_t s = 0; s < p->inputs.size(); ++s) {
        const ProgInput& in = p->inputs[s];
        if (v >= in.base && v < in.base + (uint64_t)in.bytes) {
          p->patches.push_back(ProgPatch{(unsigned)off, (int)s, (int64_t)(v - in.base)});
          break;
        }
      }
    }
  }
  // diagnostic (dt_program_info item 5): argument words that are NOT declared pointers but whose bytes happen to lie in an input
  // range -- what a scan of the argument bytes would have patched by mistake (struct padding, two adjacent 32-bit fields)
  if (!p->inputs.empty()) {
    std::set<unsigned> declared(p->ptr_off.begin(), p->ptr_off.end());
    for (size_t off = 0; off + 8 <= p->blob.size(); off += 8) {
      if (declared.count((unsigned)off)) continue;
      uint64_t v;
      memcpy(&v, p->blob.data() + off, 8);
      for (const ProgInput& in : p->inputs)
        if (v >= in.base && v < in.base + (uint64_t)in.bytes) {
          ++p->lookalikes;
          break;
        }
    }
  }
  {
    std::lock_guard<std::mutex> lock(g_prog_mutex);
    g_programs.insert(p);
  }
  *out = reinterpret_cast<dt_program_t>(p);
  return 0;
}

int dt_program_launch(dt_program_t h, int segment, const void* const* inputs, int num_inputs, dt_stream_t s) {
  Program* p = live(h);
  DT_REQUIRE(p != nullptr, "dt_program_launch: not a live program handle");
  DT_REQUIRE(to_stream(s) == p->stream,
             "dt_program_launch: a program replays on the stream it was recorded on (per-stream library scratch is baked in)");
  DT_REQUIRE(num_inputs == (int)p->inputs.size(), "dt_program_launch: %d input pointers for a program with %d input slots",
             num_inputs, (int)p->inputs.size());
  const int nseg = (int)p->seg_begin.size();
  DT_REQUIRE(segment >= -1 && segment < nseg, "dt_program_launch: segment %d of %d", segment, nseg);
  DT_REQUIRE(t_rec == nullptr || t_rec->stream != p->stream, "dt_program_launch: the stream is being recorded");
  for (int i = 0; i < num_inputs; ++i) DT_REQUIRE(inputs[i] != nullptr, "dt_program_launch: input %d is null", i);
  // (patches are applied on every call, for the whole program: ~100 8-byte stores)
  if (segment <= 0)
    for (const ProgPatch& pt : p->patches) {
      const uint64_t v = (uint64_t)reinterpret_cast<uintptr_t>(inputs[pt.slot]) + (uint64_t)pt.delta;
      memcpy(p->blob.data() + pt.blob_off, &v, 8);
    }
  const unsigned lo = segment < 0 ? 0u : p->seg_begin[segment];
  const unsigned hi = (segment < 0 || segment + 1 >= nseg) ? (unsigned)p->nodes.size() : p->seg_begin[segment + 1];
  for (unsigned i = lo; i < hi; ++i) {
    const ProgNode& n = p->nodes[i];
    note_launch();
    hipError_t e = hipLaunchKernel(n.func, n.grid, n.block, p->arg_ptrs.data() + n.arg_begin, n.shmem, p->stream);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      return fail("dt_program_launch: node %u of %zu: %s", i, p->nodes.size(), hipGetErrorString(e));
    }
  }
  return 0;
}

int64_t dt_program_info(dt_program_t h, int what) {
  Program* p = live(h);
  if (p == nullptr) {
    fail("dt_program_info: not a live program handle");
    return -1;
  }
  switch (what) {
    case 0: return (int64_t)p->nodes.size();
    case 1: return (int64_t)p->seg_begin.size();
    case 2: return (int64_t)p->patches.size();
    case 3: return (int64_t)p->inputs.size();
    case 4: return (int64_t)p->blob.size();
    case 5: return (int64_t)p->lookalikes;
    default: fail("dt_program_info: unknown item %d", what); return -1;
  }
}

int dt_program_free(dt_program_t h) {
  Program* p = reinterpret_cast<Program*>(h);
  {
    std::lock_guard<std::mutex> lock(g_prog_mutex);
    if (!g_programs.erase(p)) return fail("dt_program_free: not a live program handle");
  }
  delete p;
  return 0;
}

}  // extern "C"
