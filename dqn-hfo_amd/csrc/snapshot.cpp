// snapshot.cpp — Caffe snapshot layout for the learner (reference src/dqn.cpp:80-158, 525-620):
// `.caffemodel` / `.solverstate` written and read with a hand-rolled protobuf wire codec (no
// libprotobuf in the image), snapshot naming, old-snapshot removal, FindLatestSnapshot.
// A pure client of the C-ABI in include/dqnhip.h: no device code here.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <limits>
#include <regex>
#include <string>
#include <vector>

#include "../../include/dqnhip.h"

namespace fs = std::filesystem;

extern "C" int dqnhip_internal_set_error(const char* msg);

namespace {

int fail(const std::string& m) { dqnhip_internal_set_error(m.c_str()); return 1; }

// ---- protobuf wire format ------------------------------------------------------------------
void put_varint(std::string& o, uint64_t v) { while (v >= 0x80) { o.push_back((char)(v | 0x80)); v >>= 7; } o.push_back((char)v); }
void put_tag(std::string& o, int field, int wt) { put_varint(o, ((uint64_t)field << 3) | wt); }
void put_bytes(std::string& o, int field, const std::string& b) { put_tag(o, field, 2); put_varint(o, b.size()); o += b; }
void put_int(std::string& o, int field, int64_t v) { put_tag(o, field, 0); put_varint(o, (uint64_t)v); }

// caffe.BlobProto{shape = 7 {dim = 1 [packed]}, data = 5 [packed]}
std::string blob_proto(const std::vector<int64_t>& shape, const float* data, size_t n) {
  std::string dims;
  for (int64_t d : shape) put_varint(dims, (uint64_t)d);
  std::string bs; put_bytes(bs, 1, dims);
  std::string o;
  put_tag(o, 5, 2); put_varint(o, n * 4); o.append(reinterpret_cast<const char*>(data), n * 4);
  put_bytes(o, 7, bs);
  return o;
}

struct Reader {
  const uint8_t* p; const uint8_t* e; bool ok = true;
  uint64_t varint() { uint64_t v = 0; int s = 0; while (p < e) { uint8_t b = *p++; v |= (uint64_t)(b & 0x7f) << s; if (!(b & 0x80)) return v; s += 7; if (s > 63) break; } ok = false; return 0; }
  bool next(int& field, int& wt) { if (p >= e) return false; uint64_t t = varint(); field = (int)(t >> 3); wt = (int)(t & 7); return ok; }
  Reader sub() { uint64_t n = varint(); Reader r{p, p + n}; if (n > (uint64_t)(e - p)) { ok = false; r.e = p; } p += n; return r; }
  void skip(int wt) {
    if (wt == 0) varint(); else if (wt == 1) p += 8; else if (wt == 5) p += 4; else if (wt == 2) { uint64_t n = varint(); p += n; } else ok = false;
    if (p > e) ok = false;
  }
};

struct Blob { std::vector<int64_t> shape; std::vector<float> data; };

bool parse_blob(Reader r, Blob& b) {
  int f, wt; int64_t legacy[4] = {0, 0, 0, 0}; bool has_legacy = false;
  while (r.next(f, wt)) {
    if (f == 5 && wt == 2) { Reader d = r.sub(); size_t n = (d.e - d.p) / 4; size_t o = b.data.size(); b.data.resize(o + n); memcpy(b.data.data() + o, d.p, n * 4); }
    else if (f == 5 && wt == 5) { float v; memcpy(&v, r.p, 4); r.p += 4; b.data.push_back(v); }
    else if (f == 7 && wt == 2) { Reader s = r.sub(); int f2, w2; while (s.next(f2, w2)) { if (f2 == 1 && w2 == 2) { Reader d = s.sub(); while (d.p < d.e && d.ok) b.shape.push_back((int64_t)d.varint()); } else if (f2 == 1 && w2 == 0) b.shape.push_back((int64_t)s.varint()); else s.skip(w2); } }
    else if (f >= 1 && f <= 4 && wt == 0) { legacy[f - 1] = (int64_t)r.varint(); has_legacy = true; }
    else r.skip(wt);
    if (!r.ok) return false;
  }
  if (b.shape.empty() && has_legacy) b.shape.assign(legacy, legacy + 4);
  return r.ok;
}

struct Layer { std::string name, type; std::vector<Blob> blobs; };

bool parse_layer(Reader r, Layer& l) {
  int f, wt;
  while (r.next(f, wt)) {
    if (f == 1 && wt == 2) { Reader s = r.sub(); l.name.assign((const char*)s.p, s.e - s.p); }
    else if (f == 2 && wt == 2) { Reader s = r.sub(); l.type.assign((const char*)s.p, s.e - s.p); }
    else if (f == 7 && wt == 2) { Blob b; if (!parse_blob(r.sub(), b)) return false; l.blobs.push_back(std::move(b)); }
    else r.skip(wt);
    if (!r.ok) return false;
  }
  return true;
}

bool read_file(const std::string& path, std::string& out) {
  std::ifstream f(path, std::ios::binary);
  if (!f) return false;
  out.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
  return true;
}
bool write_file(const std::string& path, const std::string& data) {
  std::ofstream f(path, std::ios::binary | std::ios::trunc);
  if (!f) return false;
  f.write(data.data(), (std::streamsize)data.size());
  return (bool)f;
}

// ---- the learner's parameter blobs, in Caffe learnable_params order --------------------------
struct ParamBlob { std::string layer; std::vector<int64_t> shape; size_t offset, count; };

int blob_table(dqnhip_handle h, int net, std::vector<ParamBlob>& out, size_t& total) {
  dqnhip_config c;
  if (dqnhip_get_config(h, &c)) return 1;
  const bool actor = net == DQNHIP_ACTOR;
  int k = actor ? c.state_size : c.state_size + DQNHIP_ACTOR_OUT;
  size_t off = 0;
  auto add = [&](const std::string& name, int n_out, int kk) {
    out.push_back({name, {n_out, kk}, off, (size_t)n_out * kk}); off += (size_t)n_out * kk;
    out.push_back({name, {n_out}, off, (size_t)n_out}); off += n_out;
  };
  for (int i = 0; i < c.num_hidden; ++i) { add("ip" + std::to_string(i + 1) + "_layer", c.hidden[i], k); k = c.hidden[i]; }   // Tower(), src/dqn.cpp:406
  if (actor) { add("action_layer", DQNHIP_ACTION_SIZE, k); add("actionpara_layer", DQNHIP_ACTION_PARAM_SIZE, k); }          // :426-427
  else add("q_values_layer", 1, k);                                                                                         // src/dqn.hpp:43
  total = off;
  return 0;
}

// Net::ToProto (what Solver::Snapshot writes, src/dqn.cpp:589-590): the net's name, then EVERY layer of the
// initialised net with its LayerParameter as Net::Init left it — the layers CreateActorNet / CreateCriticNet build
// (src/dqn.cpp:399-455: MemoryData inputs, Silence, Concat, the Tower() of InnerProduct + in-place ReLU(0.01) pairs,
// the heads, EuclideanLoss), the Split layer InsertSplits adds where two layers consume one top (the actor's tower
// top feeds action_layer and actionpara_layer: "ip<L>_ip<L>_relu_layer_0_split"), `phase: TRAIN` on every layer
// (Net::Init sets it where absent) — and the learnable blobs on the InnerProduct layers.  Fields are emitted in
// field-number order, as the C++ protobuf serialiser does.  caffe.proto @2ef5847 numbers: LayerParameter name 1, type 2,
// bottom 3, top 4, blobs 7, phase 10, concat_param 104 {axis 2}, inner_product_param 117 {num_output 1, weight_filler 3
// {type 1, std 6}}, memory_data_param 119 {batch_size 1, channels 2, height 3, width 4}, relu_param 123 {negative_slope 1}
// (not re-verifiable offline; CopyTrainedLayersFrom only needs name + blobs, which the tests pin).
void put_float(std::string& o, int field, float v) { put_tag(o, field, 5); o.append(reinterpret_cast<const char*>(&v), 4); }

struct LayerSpec {
  std::string name, type;
  std::vector<std::string> bottom, top;
  std::string params;                       // the layer-type parameter message, already tagged (field >= 100)
  int blob0 = -1;                           // index into the ParamBlob table of this layer's weight blob (-1: no blobs)
};

std::string layer_proto(const LayerSpec& l, const std::vector<ParamBlob>& tbl, const std::vector<float>& w) {
  std::string o;
  put_bytes(o, 1, l.name);
  put_bytes(o, 2, l.type);
  for (auto& b : l.bottom) put_bytes(o, 3, b);
  for (auto& t : l.top) put_bytes(o, 4, t);
  if (l.blob0 >= 0)
    for (int k = 0; k < 2; ++k) put_bytes(o, 7, blob_proto(tbl[l.blob0 + k].shape, w.data() + tbl[l.blob0 + k].offset, tbl[l.blob0 + k].count));
  put_int(o, 10, 0);                        // phase: TRAIN
  o += l.params;
  return o;
}

std::string net_proto(const std::vector<ParamBlob>& tbl, const std::vector<float>& w, bool actor, const dqnhip_config& c) {
  auto memory_data = [&](const std::string& name, const std::string& top, const std::string& dummy, int batch, int channels, int height) {
    LayerSpec l; l.name = name; l.type = "MemoryData"; l.top = {top, dummy};
    std::string p; put_int(p, 1, batch); put_int(p, 2, channels); put_int(p, 3, height); put_int(p, 4, 1);
    put_bytes(l.params, 119, p);
    return l;
  };
  auto inner_product = [&](const std::string& name, const std::string& bottom, const std::string& top, int num_output, int blob0) {
    LayerSpec l; l.name = name; l.type = "InnerProduct"; l.bottom = {bottom}; l.top = {top}; l.blob0 = blob0;
    std::string filler; put_bytes(filler, 1, "gaussian"); put_float(filler, 6, 0.01f);
    std::string p; put_int(p, 1, num_output); put_bytes(p, 3, filler);
    put_bytes(l.params, 117, p);
    return l;
  };
  std::vector<LayerSpec> layers;
  const int B = c.minibatch, S = c.state_size;
  layers.push_back(memory_data("state_input_layer", "states", "dummy1", B, 1, S));                        // :421-422, :434-435
  std::string input = "states";
  if (actor) {
    LayerSpec sl; sl.name = "silence"; sl.type = "Silence"; sl.bottom = {"dummy1"}; layers.push_back(sl);  // :423
  } else {
    layers.push_back(memory_data("action_input_layer", "actions", "dummy2", B, 1, DQNHIP_ACTION_SIZE));    // :436-438
    layers.push_back(memory_data("action_params_input_layer", "action_params", "dummy3", B, 1, DQNHIP_ACTION_PARAM_SIZE));
    layers.push_back(memory_data("target_input_layer", "target", "dummy4", B, 1, 1));                       // :442-443 {B,1,1,1}
    LayerSpec sl; sl.name = "silence"; sl.type = "Silence"; sl.bottom = {"dummy1", "dummy2", "dummy3", "dummy4"}; layers.push_back(sl);
    LayerSpec cc; cc.name = "concat"; cc.type = "Concat"; cc.bottom = {"states", "actions", "action_params"}; cc.top = {"state_actions"};
    std::string p; put_int(p, 2, 2); put_bytes(cc.params, 104, p);                                           // axis 2, :445-447
    layers.push_back(cc);
    input = "state_actions";
  }
  int blob = 0;
  for (int i = 1; i <= c.num_hidden; ++i) {                                                                  // Tower(), :399-415
    const std::string top = "ip" + std::to_string(i);
    layers.push_back(inner_product(top + "_layer", input, top, c.hidden[i - 1], blob)); blob += 2;
    LayerSpec r; r.name = top + "_relu_layer"; r.type = "ReLU"; r.bottom = {top}; r.top = {top};
    std::string p; put_float(p, 1, 0.01f); put_bytes(r.params, 123, p);
    layers.push_back(r);
    input = top;
  }
  if (actor) {
    // InsertSplits: the tower top (last written by the in-place ReLU) has two consumers
    const std::string sp = input + "_" + input + "_relu_layer_0_split";
    LayerSpec s; s.name = sp; s.type = "Split"; s.bottom = {input}; s.top = {sp + "_0", sp + "_1"};
    layers.push_back(s);
    layers.push_back(inner_product("action_layer", sp + "_0", "actions", DQNHIP_ACTION_SIZE, blob)); blob += 2;           // :426
    layers.push_back(inner_product("actionpara_layer", sp + "_1", "action_params", DQNHIP_ACTION_PARAM_SIZE, blob));      // :427
  } else {
    layers.push_back(inner_product("q_values_layer", input, "q_values", 1, blob));                                         // :450
    LayerSpec l; l.name = "loss"; l.type = "EuclideanLoss"; l.bottom = {"q_values", "target"}; l.top = {"loss"};           // :451-452
    layers.push_back(l);
  }
  std::string o;
  put_bytes(o, 1, actor ? "Actor" : "Critic");           // np.set_name, src/dqn.cpp:420,433
  for (auto& l : layers) put_bytes(o, 100, layer_proto(l, tbl, w));
  return o;
}

int load_net_proto(const std::string& bytes, const std::vector<ParamBlob>& tbl, std::vector<float>& w, const std::string& path) {
  Reader r{(const uint8_t*)bytes.data(), (const uint8_t*)bytes.data() + bytes.size()};
  int f, wt; int matched = 0;
  while (r.next(f, wt)) {
    if (f == 100 && wt == 2) {
      Layer l;
      if (!parse_layer(r.sub(), l)) return fail(path + ": malformed LayerParameter");
      for (size_t i = 0; i < tbl.size(); i += 2) {
        if (tbl[i].layer != l.name) continue;       // CopyTrainedLayersFrom: match by name, ignore the rest
        if (l.blobs.size() != 2) return fail(path + ": layer " + l.name + " has " + std::to_string(l.blobs.size()) + " blobs, expected 2");
        for (int b = 0; b < 2; ++b) {
          if (l.blobs[b].data.size() != tbl[i + b].count)
            return fail(path + ": shape mismatch in layer " + l.name + " blob " + std::to_string(b));   // Caffe CHECKs ShapeEquals
          memcpy(w.data() + tbl[i + b].offset, l.blobs[b].data.data(), tbl[i + b].count * 4);
        }
        ++matched;
      }
    } else r.skip(wt);
    if (!r.ok) return fail(path + ": malformed NetParameter");
  }
  return matched > 0 ? 0 : fail(path + ": no layer of this net found");
}

// ---- file search helpers (src/dqn.cpp:80-121, 559-580) ------------------------------------------
std::vector<std::string> files_matching_regexp(const std::string& regexp) {
  fs::path stem(regexp), dir(fs::current_path());
  if (stem.has_parent_path()) { dir = stem.parent_path(); stem = stem.filename(); }
  std::vector<std::string> out;
  std::error_code ec;
  if (!fs::is_directory(dir, ec)) return out;
  const std::regex re(stem.string());
  for (auto& it : fs::directory_iterator(dir, ec))
    if (it.is_regular_file(ec) && std::regex_match(it.path().filename().string(), re)) out.push_back(it.path().string());
  return out;
}
int parse_iter(const std::string& s) { const size_t a = s.find_last_of('_'), b = s.find_last_of('.'); return std::stoi(s.substr(a + 1, b - a - 1)); }
int greatest_iter(const std::string& regexp) { int m = -1; for (auto& f : files_matching_regexp(regexp)) m = std::max(m, parse_iter(f)); return m; }
void remove_snapshots(const std::string& regexp, int min_iter) {
  for (auto& f : files_matching_regexp(regexp)) if (parse_iter(f) < min_iter) { std::error_code ec; fs::remove(f, ec); }
}
std::string esc(const std::string& s) {            // the reference concatenates the raw prefix into a regex; escape dots only
  return s;
}

}  // namespace

extern "C" {

int dqnhip_save_caffemodel(dqnhip_handle h, int32_t net, const char* filename) {
  if (!h || !filename) return fail("null argument");
  if (net != DQNHIP_ACTOR && net != DQNHIP_CRITIC) return fail("net must be ACTOR or CRITIC");
  std::vector<ParamBlob> tbl; size_t total = 0;
  if (blob_table(h, net, tbl, total)) return 1;
  std::vector<float> w(total);
  if (dqnhip_get_params(h, net, DQNHIP_KIND_W, w.data(), total)) return 1;
  dqnhip_config cfg;
  if (dqnhip_get_config(h, &cfg)) return 1;
  if (!write_file(filename, net_proto(tbl, w, net == DQNHIP_ACTOR, cfg))) return fail(std::string("cannot write ") + filename);
  return 0;
}

int dqnhip_load_caffemodel(dqnhip_handle h, int32_t net, const char* filename) {
  if (!h || !filename) return fail("null argument");
  if (net != DQNHIP_ACTOR && net != DQNHIP_CRITIC) return fail("net must be ACTOR or CRITIC");
  std::string bytes;
  if (!read_file(filename, bytes)) return fail(std::string("Invalid file: ") + filename);      // CHECK(is_regular_file), :526,534
  std::vector<ParamBlob> tbl; size_t total = 0;
  if (blob_table(h, net, tbl, total)) return 1;
  std::vector<float> w(total);
  if (dqnhip_get_params(h, net, DQNHIP_KIND_W, w.data(), total)) return 1;   // layers absent from the file keep their values
  if (load_net_proto(bytes, tbl, w, filename)) return 1;
  if (dqnhip_set_params(h, net, DQNHIP_KIND_W, w.data(), total)) return 1;
  return dqnhip_clone_to_target(h, net);                                      // CloneNet, :530,538
}

int dqnhip_solver_snapshot(dqnhip_handle h, int32_t net, const char* prefix, int32_t* iter_out) {
  if (!h || !prefix) return fail("null argument");
  if (net != DQNHIP_ACTOR && net != DQNHIP_CRITIC) return fail("net must be ACTOR or CRITIC");
  int32_t ia = 0, ic = 0;
  if (dqnhip_get_iters(h, &ia, &ic)) return 1;
  const int iter = net == DQNHIP_ACTOR ? ia : ic;
  const std::string base = std::string(prefix) + "_iter_" + std::to_string(iter);   // Solver::SnapshotFilename
  if (dqnhip_save_caffemodel(h, net, (base + ".caffemodel").c_str())) return 1;
  std::vector<ParamBlob> tbl; size_t total = 0;
  if (blob_table(h, net, tbl, total)) return 1;
  std::vector<float> m(total), v(total);
  if (dqnhip_get_params(h, net, DQNHIP_KIND_M, m.data(), total)) return 1;
  if (dqnhip_get_params(h, net, DQNHIP_KIND_V, v.data(), total)) return 1;
  std::string o;
  put_int(o, 1, iter);
  put_bytes(o, 2, base + ".caffemodel");
  for (auto& b : tbl) put_bytes(o, 3, blob_proto(b.shape, m.data() + b.offset, b.count));   // AdamSolver history: all m, then all v
  for (auto& b : tbl) put_bytes(o, 3, blob_proto(b.shape, v.data() + b.offset, b.count));
  put_int(o, 4, 0);                                                                           // current_step
  if (!write_file(base + ".solverstate", o)) return fail("cannot write " + base + ".solverstate");
  if (iter_out) *iter_out = iter;
  return 0;
}

int dqnhip_solver_restore(dqnhip_handle h, int32_t net, const char* solverstate) {
  if (!h || !solverstate) return fail("null argument");
  if (net != DQNHIP_ACTOR && net != DQNHIP_CRITIC) return fail("net must be ACTOR or CRITIC");
  std::string bytes;
  if (!read_file(solverstate, bytes)) return fail(std::string("Invalid file: ") + solverstate);   // :542,551
  std::vector<ParamBlob> tbl; size_t total = 0;
  if (blob_table(h, net, tbl, total)) return 1;
  Reader r{(const uint8_t*)bytes.data(), (const uint8_t*)bytes.data() + bytes.size()};
  int f, wt, iter = 0; std::string learned; std::vector<Blob> hist;
  while (r.next(f, wt)) {
    if (f == 1 && wt == 0) iter = (int)r.varint();
    else if (f == 2 && wt == 2) { Reader s = r.sub(); learned.assign((const char*)s.p, s.e - s.p); }
    else if (f == 3 && wt == 2) { Blob b; if (!parse_blob(r.sub(), b)) return fail(std::string(solverstate) + ": malformed history blob"); hist.push_back(std::move(b)); }
    else r.skip(wt);
    if (!r.ok) return fail(std::string(solverstate) + ": malformed SolverState");
  }
  if (hist.size() != 2 * tbl.size()) return fail(std::string(solverstate) + ": Incorrect length of history blobs.");   // Caffe CHECK_EQ
  if (!learned.empty()) {
    // Solver::Restore -> net_->CopyTrainedLayersFrom(learned_net); a relative or stale path is
    // retried next to the .solverstate (snapshots are renamed after writing, src/dqn.cpp:596-604)
    std::string path = learned;
    std::error_code ec;
    if (!fs::is_regular_file(path, ec)) {
      std::string alt = solverstate; const size_t dot = alt.rfind(".solverstate");
      if (dot != std::string::npos) alt = alt.substr(0, dot) + ".caffemodel";
      path = alt;
    }
    if (dqnhip_load_caffemodel(h, net, path.c_str())) return 1;
  }
  std::vector<float> m(total), v(total);
  for (size_t i = 0; i < tbl.size(); ++i) {
    if (hist[i].data.size() != tbl[i].count || hist[i + tbl.size()].data.size() != tbl[i].count)
      return fail(std::string(solverstate) + ": history blob shape mismatch");
    memcpy(m.data() + tbl[i].offset, hist[i].data.data(), tbl[i].count * 4);
    memcpy(v.data() + tbl[i].offset, hist[i + tbl.size()].data.data(), tbl[i].count * 4);
  }
  if (dqnhip_set_params(h, net, DQNHIP_KIND_M, m.data(), total)) return 1;
  if (dqnhip_set_params(h, net, DQNHIP_KIND_V, v.data(), total)) return 1;
  int32_t ia = 0, ic = 0;
  if (dqnhip_get_iters(h, &ia, &ic)) return 1;
  if (net == DQNHIP_ACTOR) ia = iter; else ic = iter;
  if (dqnhip_set_iters(h, ia, ic)) return 1;
  return dqnhip_clone_to_target(h, net);          // targets are NOT checkpointed: re-cloned on restore (:546,555)
}

int dqnhip_snapshot(dqnhip_handle h, const char* save_path, const char* snapshot_prefix, int32_t remove_old,
                    int32_t snapshot_memory) {
  if (!h || !save_path || !snapshot_prefix) return fail("null argument");
  const std::string sp(save_path), pre(snapshot_prefix);
  int32_t actor_iter = 0, critic_iter = 0;
  if (dqnhip_solver_snapshot(h, DQNHIP_ACTOR, (sp + "_actor").c_str(), &actor_iter)) return 1;    // snapshot_prefix of the solvers, src/dqn_main.cpp:247-248
  if (dqnhip_solver_snapshot(h, DQNHIP_CRITIC, (sp + "_critic").c_str(), &critic_iter)) return 1;
  std::error_code ec;
  auto mv = [&](const std::string& a, const std::string& b) -> int {
    if (!fs::is_regular_file(a, ec)) return fail("missing " + a);                                  // CHECK(is_regular_file), :593-601
    if (a != b) fs::rename(a, b, ec);
    return ec ? fail("rename " + a + " -> " + b + " failed") : 0;
  };
  const std::string af = sp + "_actor_iter_" + std::to_string(actor_iter), at = pre + "_actor_iter_" + std::to_string(actor_iter);
  const std::string cf = sp + "_critic_iter_" + std::to_string(critic_iter), ct = pre + "_critic_iter_" + std::to_string(critic_iter);
  if (mv(af + ".caffemodel", at + ".caffemodel") || mv(af + ".solverstate", at + ".solverstate")) return 1;
  if (mv(cf + ".caffemodel", ct + ".caffemodel") || mv(cf + ".solverstate", ct + ".solverstate")) return 1;
  if (snapshot_memory) {
    const std::string mem = pre + "_iter_" + std::to_string(std::max(actor_iter, critic_iter)) + ".replaymemory";
    if (dqnhip_snapshot_replay_memory(h, mem.c_str())) return 1;
    if (!fs::is_regular_file(mem, ec)) return fail("missing " + mem);
  }
  if (remove_old) {                                                                                // :612-618
    remove_snapshots(esc(pre) + "_actor_iter_[0-9]+\\.(caffemodel|solverstate)", actor_iter - 1);
    remove_snapshots(esc(pre) + "_critic_iter_[0-9]+\\.(caffemodel|solverstate)", critic_iter - 1);
    remove_snapshots(esc(pre) + "_iter_[0-9]+\\.replaymemory", critic_iter - 1);
  }
  return 0;
}

int dqnhip_find_latest_snapshot(const char* snapshot_prefix, char* actor, char* critic, char* memory, size_t buf_len) {
  if (!snapshot_prefix || !actor || !critic || !memory || buf_len == 0) return fail("null argument");
  const std::string pre(snapshot_prefix);
  actor[0] = critic[0] = memory[0] = 0;
  const int a = greatest_iter(esc(pre) + "_actor_iter_[0-9]+\\.solverstate");
  const int c = greatest_iter(esc(pre) + "_critic_iter_[0-9]+\\.solverstate");
  const int m = greatest_iter(esc(pre) + "_iter_[0-9]+\\.replaymemory");
  if (a > 0) snprintf(actor, buf_len, "%s_actor_iter_%d.solverstate", pre.c_str(), a);
  if (c > 0) snprintf(critic, buf_len, "%s_critic_iter_%d.solverstate", pre.c_str(), c);
  if (m > 0) snprintf(memory, buf_len, "%s_iter_%d.replaymemory", pre.c_str(), m);
  return 0;
}

int dqnhip_find_hiscore(const char* snapshot_prefix, int32_t* score) {
  if (!snapshot_prefix || !score) return fail("null argument");
  int best = std::numeric_limits<int>::lowest();
  for (auto& f : files_matching_regexp(esc(snapshot_prefix) + "_HiScore[-]?[0-9]+_iter_[0-9]+\\.caffemodel")) {
    const size_t a = f.find("_HiScore"), b = f.find("_iter_");
    best = std::max(best, std::stoi(f.substr(a + 8, b - a - 1)));                                   // ParseScoreFromSnapshot, :86-90
  }
  *score = best;
  return 0;
}

int dqnhip_remove_files_matching_regexp(const char* regexp) {
  if (!regexp) return fail("null argument");
  for (auto& f : files_matching_regexp(regexp)) { std::error_code ec; fs::remove(f, ec); }
  return 0;
}

// FilesMatchingRegexp (src/dqn.hpp:213-216, src/dqn.cpp:559-580): the regular files of the regexp's
// directory whose NAME matches its last path component, '\n'-separated into buf.
int dqnhip_files_matching_regexp(const char* regexp, char* buf, size_t buf_len, int32_t* count) {
  if (!regexp || !buf || buf_len == 0) return fail("null argument");
  std::vector<std::string> files;
  try { files = files_matching_regexp(regexp); }
  catch (const std::regex_error& e) { return fail(std::string("bad regexp '") + regexp + "': " + e.what()); }
  std::sort(files.begin(), files.end());       // directory order is unspecified; make the listing stable
  std::string joined;
  for (auto& f : files) { if (!joined.empty()) joined += '\n'; joined += f; }
  if (joined.size() + 1 > buf_len) return fail("files_matching_regexp: buffer too small (" + std::to_string(joined.size() + 1) + " bytes needed)");
  memcpy(buf, joined.c_str(), joined.size() + 1);
  if (count) *count = (int32_t)files.size();
  return 0;
}

// RemoveSnapshots (src/dqn.hpp:222-224, src/dqn.cpp:100-109): remove the matching files whose
// "_<iter>." suffix is below min_iter.
int dqnhip_remove_snapshots(const char* regexp, int32_t min_iter) {
  if (!regexp) return fail("null argument");
  try { remove_snapshots(regexp, min_iter); }
  catch (const std::exception& e) { return fail(std::string("RemoveSnapshots(") + regexp + "): " + e.what()); }
  return 0;
}

}  // extern "C"
