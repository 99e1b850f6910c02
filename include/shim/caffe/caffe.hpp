// include/shim/caffe/caffe.hpp — the slice of BVLC Caffe's C++ API that the reference's DRIVER and
// class declaration touch (src/dqn_main.cpp:208-262, src/dqn.hpp:10, 34-35, 57-60, 116-117,
// 204-205), for boxes without Caffe (this image): caffe::Caffe::set_mode, SolverParameter with
// the protobuf-style accessors the driver calls, NetParameter / LayerParameter with the fields
// CreateActorNet / CreateCriticNet emit (src/dqn.cpp:225-454), Read/WriteProto*TextFile in
// Caffe's prototxt text format, and opaque Net / Solver / Layer / Blob templates so that the
// reference's own dqn.hpp parses.  None of Caffe's arithmetic is here: that is what
// libdqnhip.so replaces.
#ifndef DQNHIP_SHIM_CAFFE_HPP_
#define DQNHIP_SHIM_CAFFE_HPP_

#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <memory>
#include <sstream>
#include <string>
#include <vector>
// the real caffe.hpp pulls these in transitively (caffe/common.hpp, boost, glog); the reference's
// sources rely on that (std::deque src/dqn.hpp:187, sleep() src/dqn_main.cpp:425, sqrt() :166)
#include <climits>
#include <cmath>
#include <deque>
#include <iostream>
#include <map>
#include <set>
#include <utility>
#include <unistd.h>

namespace boost {       // Caffe's headers bring boost::shared_ptr into scope (src/dqn.hpp:35 NetSp)
using std::shared_ptr;
}

namespace caffe {

class Caffe {
 public:
  enum Brew { CPU, GPU };
  static void set_mode(Brew mode) { mode_() = mode; }
  static Brew mode() { return mode_(); }
 private:
  static Brew& mode_() { static thread_local Brew m = CPU; return m; }     // per thread, as in Caffe (src/dqn_main.cpp:208-212)
};

enum Phase { TRAIN = 0, TEST = 1 };

struct InnerProductParameter {
  int num_output_ = 0; std::string filler_type_ = "gaussian"; float filler_std_ = 0.01f;
  int num_output() const { return num_output_; } void set_num_output(int v) { num_output_ = v; }
};
struct ReLUParameter { float negative_slope_ = 0; float negative_slope() const { return negative_slope_; } void set_negative_slope(float v) { negative_slope_ = v; } };
struct ConcatParameter { int axis_ = 1; int axis() const { return axis_; } void set_axis(int v) { axis_ = v; } };
struct MemoryDataParameter {
  int batch_size_ = 0, channels_ = 0, height_ = 0, width_ = 0;
  int batch_size() const { return batch_size_; } int channels() const { return channels_; } int height() const { return height_; } int width() const { return width_; }
  void set_batch_size(int v) { batch_size_ = v; } void set_channels(int v) { channels_ = v; } void set_height(int v) { height_ = v; } void set_width(int v) { width_ = v; }
};

class LayerParameter {
 public:
  const std::string& name() const { return name_; } void set_name(const std::string& v) { name_ = v; }
  const std::string& type() const { return type_; } void set_type(const std::string& v) { type_ = v; }
  int bottom_size() const { return (int)bottom_.size(); } const std::string& bottom(int i) const { return bottom_[i]; } void add_bottom(const std::string& v) { bottom_.push_back(v); }
  int top_size() const { return (int)top_.size(); } const std::string& top(int i) const { return top_[i]; } void add_top(const std::string& v) { top_.push_back(v); }
  const InnerProductParameter& inner_product_param() const { return ip_; } InnerProductParameter* mutable_inner_product_param() { has_ip_ = true; return &ip_; }
  const ReLUParameter& relu_param() const { return relu_; } ReLUParameter* mutable_relu_param() { has_relu_ = true; return &relu_; }
  const ConcatParameter& concat_param() const { return concat_; } ConcatParameter* mutable_concat_param() { has_concat_ = true; return &concat_; }
  const MemoryDataParameter& memory_data_param() const { return md_; } MemoryDataParameter* mutable_memory_data_param() { has_md_ = true; return &md_; }
  bool has_inner_product_param() const { return has_ip_; } bool has_relu_param() const { return has_relu_; }
  bool has_concat_param() const { return has_concat_; } bool has_memory_data_param() const { return has_md_; }
 private:
  std::string name_, type_;
  std::vector<std::string> bottom_, top_;
  InnerProductParameter ip_; ReLUParameter relu_; ConcatParameter concat_; MemoryDataParameter md_;
  bool has_ip_ = false, has_relu_ = false, has_concat_ = false, has_md_ = false;
};

class NetParameter {
 public:
  const std::string& name() const { return name_; } void set_name(const std::string& v) { name_ = v; }
  bool force_backward() const { return force_backward_; } void set_force_backward(bool v) { force_backward_ = v; }
  int layer_size() const { return (int)layer_.size(); }
  const LayerParameter& layer(int i) const { return layer_[i]; }
  LayerParameter* mutable_layer(int i) { return &layer_[i]; }
  LayerParameter* add_layer() { layer_.emplace_back(); return &layer_.back(); }
  void CopyFrom(const NetParameter& o) { *this = o; }
  void Clear() { *this = NetParameter(); }
 private:
  std::string name_; bool force_backward_ = false;
  std::vector<LayerParameter> layer_;
};

class SolverParameter {
 public:
  NetParameter* mutable_net_param() { return &net_param_; }
  const NetParameter& net_param() const { return net_param_; }
#define DQNHIP_SHIM_FIELD(type, name, dflt)                  \
 public:                                                     \
  const type& name() const { return name##_; }               \
  void set_##name(const type& v) { name##_ = v; }            \
 private:                                                    \
  type name##_ = dflt;
  DQNHIP_SHIM_FIELD(std::string, snapshot_prefix, "")
  DQNHIP_SHIM_FIELD(std::string, type, "SGD")
  DQNHIP_SHIM_FIELD(std::string, lr_policy, "")
  DQNHIP_SHIM_FIELD(int, max_iter, 0)
  DQNHIP_SHIM_FIELD(float, base_lr, 0.0f)
  DQNHIP_SHIM_FIELD(float, momentum, 0.0f)
  DQNHIP_SHIM_FIELD(float, momentum2, 0.999f)          // caffe.proto defaults
  DQNHIP_SHIM_FIELD(float, delta, 1e-8f)
  DQNHIP_SHIM_FIELD(float, clip_gradients, -1.0f)
#undef DQNHIP_SHIM_FIELD
 private:
  NetParameter net_param_;
};

// ---- prototxt (protobuf text format) for the fields above --------------------------------------
namespace shim {
inline std::string quote(const std::string& s) { return "\"" + s + "\""; }

inline std::string to_text(const NetParameter& np) {
  std::ostringstream o;
  o << "name: " << quote(np.name()) << "\n";
  for (int i = 0; i < np.layer_size(); ++i) {
    const LayerParameter& l = np.layer(i);
    o << "layer {\n  name: " << quote(l.name()) << "\n  type: " << quote(l.type()) << "\n";
    for (int b = 0; b < l.bottom_size(); ++b) o << "  bottom: " << quote(l.bottom(b)) << "\n";
    for (int t = 0; t < l.top_size(); ++t) o << "  top: " << quote(l.top(t)) << "\n";
    if (l.has_relu_param()) o << "  relu_param {\n    negative_slope: " << l.relu_param().negative_slope() << "\n  }\n";
    if (l.has_inner_product_param())
      o << "  inner_product_param {\n    num_output: " << l.inner_product_param().num_output()
        << "\n    weight_filler {\n      type: " << quote(l.inner_product_param().filler_type_) << "\n      std: "
        << l.inner_product_param().filler_std_ << "\n    }\n  }\n";
    if (l.has_concat_param()) o << "  concat_param {\n    axis: " << l.concat_param().axis() << "\n  }\n";
    if (l.has_memory_data_param())
      o << "  memory_data_param {\n    batch_size: " << l.memory_data_param().batch_size() << "\n    channels: " << l.memory_data_param().channels()
        << "\n    height: " << l.memory_data_param().height() << "\n    width: " << l.memory_data_param().width() << "\n  }\n";
    o << "}\n";
  }
  if (np.force_backward()) o << "force_backward: true\n";
  return o.str();
}

// tokens: identifiers / numbers, quoted strings, '{', '}', ':'; '#' comments
struct Lexer {
  const std::string& s; size_t p = 0;
  explicit Lexer(const std::string& str) : s(str) {}
  bool next(std::string& tok, bool& quoted) {
    quoted = false;
    for (;;) {
      while (p < s.size() && std::isspace((unsigned char)s[p])) ++p;
      if (p < s.size() && s[p] == '#') { while (p < s.size() && s[p] != '\n') ++p; continue; }
      break;
    }
    if (p >= s.size()) return false;
    const char c = s[p];
    if (c == '{' || c == '}' || c == ':') { tok = std::string(1, c); ++p; return true; }
    if (c == '"' || c == '\'') {
      const char q = c; ++p; tok.clear(); quoted = true;
      while (p < s.size() && s[p] != q) { if (s[p] == '\\' && p + 1 < s.size()) ++p; tok += s[p++]; }
      ++p; return true;
    }
    tok.clear();
    while (p < s.size() && !std::isspace((unsigned char)s[p]) && s[p] != '{' && s[p] != '}' && s[p] != ':' && s[p] != '#') tok += s[p++];
    return true;
  }
};

// message body -> callback(path of enclosing field names, key, value); unknown fields are skipped
template <class F>
bool parse_body(Lexer& lx, std::vector<std::string>& path, F&& on_field, bool top) {
  std::string tok; bool q;
  while (lx.next(tok, q)) {
    if (tok == "}" && !q) return !top;
    const std::string key = tok;
    if (!lx.next(tok, q)) return false;
    if (tok == ":" && !q) { if (!lx.next(tok, q)) return false; }
    if (tok == "{" && !q) {
      path.push_back(key);
      on_field(path, std::string("{"), std::string());
      if (!parse_body(lx, path, on_field, false)) return false;
      on_field(path, std::string("}"), std::string());
      path.pop_back();
    } else on_field(path, key, tok);
  }
  return top;
}

inline bool from_text(const std::string& text, NetParameter* np) {
  np->Clear();
  Lexer lx(text);
  std::vector<std::string> path;
  LayerParameter* cur = nullptr;
  auto on = [&](const std::vector<std::string>& p, const std::string& k, const std::string& v) {
    if (p.empty()) { if (k == "name") np->set_name(v); else if (k == "force_backward") np->set_force_backward(v == "true"); return; }
    if (p[0] != "layer" && p[0] != "layers") return;
    if (p.size() == 1) {
      if (k == "{") cur = np->add_layer();
      else if (k == "}") cur = nullptr;
      else if (!cur) return;
      else if (k == "name") cur->set_name(v);
      else if (k == "type") cur->set_type(v);
      else if (k == "bottom") cur->add_bottom(v);
      else if (k == "top") cur->add_top(v);
      return;
    }
    if (!cur || k == "{" || k == "}") {
      if (cur && k == "{" && p.size() == 2) {
        if (p[1] == "inner_product_param") cur->mutable_inner_product_param();
        else if (p[1] == "relu_param") cur->mutable_relu_param();
        else if (p[1] == "concat_param") cur->mutable_concat_param();
        else if (p[1] == "memory_data_param") cur->mutable_memory_data_param();
      }
      return;
    }
    if (p[1] == "inner_product_param") {
      if (p.size() == 2 && k == "num_output") cur->mutable_inner_product_param()->set_num_output(std::atoi(v.c_str()));
      else if (p.size() == 3 && p[2] == "weight_filler" && k == "type") cur->mutable_inner_product_param()->filler_type_ = v;
      else if (p.size() == 3 && p[2] == "weight_filler" && k == "std") cur->mutable_inner_product_param()->filler_std_ = (float)std::atof(v.c_str());
    } else if (p[1] == "relu_param" && k == "negative_slope") cur->mutable_relu_param()->set_negative_slope((float)std::atof(v.c_str()));
    else if (p[1] == "concat_param" && k == "axis") cur->mutable_concat_param()->set_axis(std::atoi(v.c_str()));
    else if (p[1] == "memory_data_param") {
      MemoryDataParameter* m = cur->mutable_memory_data_param();
      const int x = std::atoi(v.c_str());
      if (k == "batch_size") m->set_batch_size(x); else if (k == "channels") m->set_channels(x);
      else if (k == "height") m->set_height(x); else if (k == "width") m->set_width(x);
    }
  };
  return parse_body(lx, path, on, true);
}
}  // namespace shim

inline bool ReadProtoFromTextFile(const char* filename, NetParameter* proto) {
  std::ifstream f(filename);
  if (!f) return false;
  std::stringstream ss; ss << f.rdbuf();
  return shim::from_text(ss.str(), proto);
}
inline void ReadProtoFromTextFileOrDie(const char* filename, NetParameter* proto) {
  if (!ReadProtoFromTextFile(filename, proto)) { std::fprintf(stderr, "F caffe shim: Check failed: ReadProtoFromTextFile(%s)\n", filename); std::abort(); }
}
inline void ReadProtoFromTextFileOrDie(const std::string& filename, NetParameter* proto) { ReadProtoFromTextFileOrDie(filename.c_str(), proto); }
inline void WriteProtoToTextFile(const NetParameter& proto, const char* filename) {
  std::ofstream f(filename);
  f << shim::to_text(proto);
  if (!f) { std::fprintf(stderr, "F caffe shim: cannot write %s\n", filename); std::abort(); }
}
inline void WriteProtoToTextFile(const NetParameter& proto, const std::string& filename) { WriteProtoToTextFile(proto, filename.c_str()); }

// opaque: only so that declarations mentioning them (src/dqn.hpp:34-35, 116-117, 149-184) parse
template <typename Dtype> class Blob;
template <typename Dtype> class Layer;
template <typename Dtype> class Net;
template <typename Dtype> class Solver { public: int iter() const { return iter_; } private: int iter_ = 0; };

}  // namespace caffe

#endif  // DQNHIP_SHIM_CAFFE_HPP_
