// proto.h -- reader/writer for the reference's model files: protobuf (proto2) messages of
// clstm.proto:1-26, produced/consumed by clstm_proto.cc:35-180.  libprotobuf is not available in this
// image, so the wire format is implemented directly (varints, length-delimited fields, fixed32
// floats); files written here are byte-for-byte what SerializeToOstream emits for the same content
// (fields in number order, proto2 repeated scalars UNPACKED) and the reader also accepts packed
// repeated fields.  tests/test_host_tools.py cross-checks both directions against python-protobuf.
//
//   KeyValue     { 1 key, 2 value }
//   Array        { 1 name, 2 dim*, 3 value* }           values row-major (clstm_proto.cc:43-44)
//   NetworkProto { 1 kind, 2 name, 10 ninput, 11 noutput, 12 icodec*, 13 codec*,
//                  20 attribute*, 30 weights*, 40 sub* }
#pragma once
#include <map>
#include <memory>

#include "hostutil.h"

namespace clstmhost {

struct ArrayProto {
  string name;
  vector<int> dim;
  vector<float> value;
};
struct NetProto {
  string kind, name;
  bool has_name = false;
  int ninput = 0, noutput = 0;
  vector<int> icodec, codec;
  vector<std::pair<string, string>> attribute;
  vector<ArrayProto> weights;
  vector<NetProto> sub;
  string attr(const string& key, const string& dflt = "") const {
    for (auto& kv : attribute)
      if (kv.first == key) return kv.second;
    return dflt;
  }
};

namespace wire {
inline void put_varint(string& out, unsigned long long v) {
  while (v >= 0x80) { out.push_back((char)((v & 0x7f) | 0x80)); v >>= 7; }
  out.push_back((char)v);
}
inline void put_tag(string& out, int field, int wt) { put_varint(out, ((unsigned long long)field << 3) | wt); }
inline void put_int32(string& out, int field, int v) {  // negative int32 is sign-extended to 10 bytes
  put_tag(out, field, 0);
  put_varint(out, (unsigned long long)(long long)v);
}
inline void put_bytes(string& out, int field, const string& s) {
  put_tag(out, field, 2);
  put_varint(out, s.size());
  out += s;
}
inline void put_float(string& out, int field, float f) {
  put_tag(out, field, 5);
  char b[4];
  memcpy(b, &f, 4);  // little-endian host
  out.append(b, 4);
}
struct Reader {
  const unsigned char* p;
  const unsigned char* end;
  bool done() const { return p >= end; }
  unsigned long long varint() {
    unsigned long long v = 0;
    int shift = 0;
    for (;;) {
      if (p >= end) fail("truncated varint in model file");
      unsigned char b = *p++;
      v |= (unsigned long long)(b & 0x7f) << shift;
      if (!(b & 0x80)) break;
      shift += 7;
      if (shift > 63) fail("bad varint in model file");
    }
    return v;
  }
  Reader sub() {
    unsigned long long n = varint();
    if ((unsigned long long)(end - p) < n) fail("truncated field in model file");
    Reader r{p, p + n};
    p += n;
    return r;
  }
  float fixed32() {
    if (end - p < 4) fail("truncated float in model file");
    float f;
    memcpy(&f, p, 4);
    p += 4;
    return f;
  }
  void skip(int wt) {
    switch (wt) {
      case 0: varint(); break;
      case 1: if (end - p < 8) fail("truncated"); p += 8; break;
      case 2: sub(); break;
      case 5: if (end - p < 4) fail("truncated"); p += 4; break;
      default: fail("unsupported wire type in model file");
    }
  }
  string str() { Reader r = sub(); return string((const char*)r.p, r.end - r.p); }
};
}  // namespace wire

inline string serialize(const ArrayProto& a) {
  string out;
  wire::put_bytes(out, 1, a.name);
  for (int d : a.dim) wire::put_int32(out, 2, d);
  for (float v : a.value) wire::put_float(out, 3, v);
  return out;
}
inline string serialize(const NetProto& n) {
  string out;
  wire::put_bytes(out, 1, n.kind);
  if (n.has_name) wire::put_bytes(out, 2, n.name);
  wire::put_int32(out, 10, n.ninput);
  wire::put_int32(out, 11, n.noutput);
  for (int c : n.icodec) wire::put_int32(out, 12, c);
  for (int c : n.codec) wire::put_int32(out, 13, c);
  for (auto& kv : n.attribute) {
    string m;
    wire::put_bytes(m, 1, kv.first);
    wire::put_bytes(m, 2, kv.second);
    wire::put_bytes(out, 20, m);
  }
  for (auto& w : n.weights) wire::put_bytes(out, 30, serialize(w));
  for (auto& s : n.sub) wire::put_bytes(out, 40, serialize(s));
  return out;
}

inline void parse_ints(wire::Reader& r, int wt, vector<int>& out) {
  if (wt == 0) out.push_back((int)r.varint());
  else if (wt == 2) {  // packed
    wire::Reader s = r.sub();
    while (!s.done()) out.push_back((int)s.varint());
  } else fail("bad wire type for int32 field");
}
inline ArrayProto parse_array(wire::Reader r) {
  ArrayProto a;
  while (!r.done()) {
    unsigned long long tag = r.varint();
    int field = (int)(tag >> 3), wt = (int)(tag & 7);
    if (field == 1 && wt == 2) a.name = r.str();
    else if (field == 2) parse_ints(r, wt, a.dim);
    else if (field == 3 && wt == 5) a.value.push_back(r.fixed32());
    else if (field == 3 && wt == 2) {
      wire::Reader s = r.sub();
      while (!s.done()) a.value.push_back(s.fixed32());
    } else r.skip(wt);
  }
  return a;
}
inline NetProto parse_net(wire::Reader r) {
  NetProto n;
  bool has_kind = false;
  while (!r.done()) {
    unsigned long long tag = r.varint();
    int field = (int)(tag >> 3), wt = (int)(tag & 7);
    if (field == 1 && wt == 2) { n.kind = r.str(); has_kind = true; }
    else if (field == 2 && wt == 2) { n.name = r.str(); n.has_name = true; }
    else if (field == 10 && wt == 0) n.ninput = (int)r.varint();
    else if (field == 11 && wt == 0) n.noutput = (int)r.varint();
    else if (field == 12) parse_ints(r, wt, n.icodec);
    else if (field == 13) parse_ints(r, wt, n.codec);
    else if (field == 20 && wt == 2) {
      wire::Reader k = r.sub();
      string key, value;
      while (!k.done()) {
        unsigned long long t = k.varint();
        if ((t >> 3) == 1 && (t & 7) == 2) key = k.str();
        else if ((t >> 3) == 2 && (t & 7) == 2) value = k.str();
        else k.skip((int)(t & 7));
      }
      n.attribute.emplace_back(key, value);
    } else if (field == 30 && wt == 2) n.weights.push_back(parse_array(r.sub()));
    else if (field == 40 && wt == 2) n.sub.push_back(parse_net(r.sub()));
    else r.skip(wt);
  }
  if (!has_kind) fail("model file: NetworkProto without kind (required field)");
  return n;
}
inline NetProto load_proto(const string& fname) {  // load_as_proto, clstm_proto.cc:173-178
  std::ifstream stream(fname, std::ios::binary);
  if (!stream) fail("cannot open: " + fname);
  string data((std::istreambuf_iterator<char>(stream)), std::istreambuf_iterator<char>());
  wire::Reader r{(const unsigned char*)data.data(), (const unsigned char*)data.data() + data.size()};
  return parse_net(r);
}
inline void save_proto(const string& fname, const NetProto& n) {  // save_as_proto, :153-157
  std::ofstream stream(fname, std::ios::binary);
  if (!stream) fail("cannot write: " + fname);
  string data = serialize(n);
  stream.write(data.data(), data.size());
}

}  // namespace clstmhost
