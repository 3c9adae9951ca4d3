// tor_mp4.cpp -- the last step of the animation driver (trace_of_radiance_animation.nim:203-210):
// wrap the Annex-B H.264 stream into an MP4 file.  The reference does this with the vendored minimp4
// (io/mp4.nim:113-163: MP4Muxer.initialize / writeMP4_from / close, 30 frames per second,
// 90 kHz time stamps, one sample per slice NAL unit); this is an independent, minimal ISO-BMFF
// writer for the same job: ftyp, mdat (4-byte length-prefixed NAL units), moov with one avc1 track.
// Host code only; streams file to file (a 1080p I_PCM stream is ~3 MB per frame).
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/tor_render.h"

extern "C" const char* tor_last_error(void);

namespace tor {
void set_last_error(const std::string& msg);  // tor_api.cpp
}

namespace {

struct Box {  // big-endian byte buffer with nested box sizes
  std::vector<uint8_t> b;
  void u8(uint32_t v) { b.push_back((uint8_t)v); }
  void u16(uint32_t v) { u8(v >> 8); u8(v); }
  void u32(uint32_t v) { u16(v >> 16); u16(v); }
  void u64(uint64_t v) { u32((uint32_t)(v >> 32)); u32((uint32_t)v); }
  void tag(const char* t) { b.insert(b.end(), t, t + 4); }
  void zeros(int n) { b.insert(b.end(), (size_t)n, 0); }
  void bytes(const uint8_t* p, size_t n) { b.insert(b.end(), p, p + n); }
  size_t open(const char* t) {  // returns the position of the size field
    const size_t at = b.size();
    u32(0);
    tag(t);
    return at;
  }
  size_t open_full(const char* t, uint32_t version, uint32_t flags) {
    const size_t at = open(t);
    u32((version << 24) | flags);
    return at;
  }
  void close(size_t at) {
    const uint32_t n = (uint32_t)(b.size() - at);
    b[at] = (uint8_t)(n >> 24); b[at + 1] = (uint8_t)(n >> 16); b[at + 2] = (uint8_t)(n >> 8); b[at + 3] = (uint8_t)n;
  }
};

void unity_matrix(Box& x) {
  const uint32_t m[9] = {0x00010000, 0, 0, 0, 0x00010000, 0, 0, 0, 0x40000000};
  for (uint32_t v : m) x.u32(v);
}

// Incremental Annex-B reader: yields the NAL units (without start codes) of a byte stream file.
struct NalReader {
  FILE* f;
  std::vector<uint8_t> buf;
  size_t pos = 0;  // first unconsumed byte
  bool eof = false;
  explicit NalReader(FILE* file) : f(file) {}
  bool more() {
    if (eof) return false;
    const size_t old = buf.size();
    buf.resize(old + (1u << 20));
    const size_t got = fread(buf.data() + old, 1, 1u << 20, f);
    buf.resize(old + got);
    if (got == 0) eof = true;
    return got > 0;
  }
  // position of the first start code (00 00 01) at or after `from`; reads on while none is buffered
  size_t find_start(size_t from) {
    size_t i = from;
    for (;;) {
      for (; i + 3 <= buf.size(); ++i)
        if (buf[i] == 0 && buf[i + 1] == 0 && buf[i + 2] == 1) return i;
      if (!more()) return (size_t)-1;
    }
  }
  bool next(std::vector<uint8_t>& nal) {
    if (pos > 0) {  // drop what the previous call consumed
      buf.erase(buf.begin(), buf.begin() + (long)pos);
      pos = 0;
    }
    const size_t s = find_start(0);
    if (s == (size_t)-1) return false;
    const size_t begin = s + 3;
    size_t end = find_start(begin);
    pos = end == (size_t)-1 ? buf.size() : end;
    if (end == (size_t)-1) end = buf.size();
    // trailing zero bytes belong to the byte-stream framing (the next 4-byte start code), not to the unit
    while (end > begin && buf[end - 1] == 0) --end;
    nal.assign(buf.begin() + (long)begin, buf.begin() + (long)end);
    return true;
  }
};

int fail(const std::string& msg) {
  tor::set_last_error(msg);
  return TOR_ERR_INVALID_ARGUMENT;
}

}  // namespace

extern "C" int tor_mp4_mux_file(const char* src_annexb_path, const char* dst_mp4_path, int32_t width, int32_t height,
                                int32_t fps) {
  if (!src_annexb_path || !dst_mp4_path || width <= 0 || height <= 0 || width > 65535 || height > 65535 || fps <= 0 ||
      fps > 90000)
    return fail("tor_mp4_mux_file: bad argument");
  FILE* in = fopen(src_annexb_path, "rb");
  if (!in) return fail(std::string("tor_mp4_mux_file: cannot open ") + src_annexb_path);
  FILE* out = fopen(dst_mp4_path, "wb");
  if (!out) {
    fclose(in);
    return fail(std::string("tor_mp4_mux_file: cannot create ") + dst_mp4_path);
  }
  bool io_ok = true;
  auto put = [&](const void* p, size_t n) { io_ok = io_ok && fwrite(p, 1, n, out) == n; };

  // ---- ftyp + mdat header (64-bit size, patched at the end) ----
  Box head;
  size_t at = head.open("ftyp");
  head.tag("isom"); head.u32(0x200); head.tag("isom"); head.tag("iso2"); head.tag("avc1"); head.tag("mp41");
  head.close(at);
  const uint64_t mdat_at = head.b.size();
  head.u32(1);  // size == 1: 64-bit largesize follows
  head.tag("mdat");
  head.u64(0);
  put(head.b.data(), head.b.size());
  uint64_t offset = head.b.size();  // file offset of the next sample

  // ---- samples ----
  std::vector<uint8_t> sps, pps, nal, pending;  // pending: SEI / AUD units that precede a slice
  std::vector<uint32_t> sizes;
  std::vector<uint64_t> offsets;
  std::vector<uint32_t> sync;  // 1-based sample numbers of IDR pictures
  NalReader rd(in);
  std::string err;
  while (err.empty() && rd.next(nal)) {
    if (nal.empty()) continue;
    const int type = nal[0] & 0x1f;
    if (type == 7 || type == 8) {
      std::vector<uint8_t>& dst = type == 7 ? sps : pps;
      if (dst.empty()) dst = nal;
      else if (dst != nal) err = "tor_mp4_mux_file: the stream changes its SPS/PPS (one decoder configuration per track)";
    } else if (type == 1 || type == 5) {
      const uint64_t sample_at = offset;
      uint32_t sample_bytes = 0;
      auto put_nal = [&](const std::vector<uint8_t>& u) {
        const uint8_t len[4] = {(uint8_t)(u.size() >> 24), (uint8_t)(u.size() >> 16), (uint8_t)(u.size() >> 8), (uint8_t)u.size()};
        put(len, 4);
        put(u.data(), u.size());
        sample_bytes += 4 + (uint32_t)u.size();
      };
      if (!pending.empty()) {
        // pending holds length-prefixed units already
        put(pending.data(), pending.size());
        sample_bytes += (uint32_t)pending.size();
        pending.clear();
      }
      if (nal.size() > 0x7fffffffu) { err = "tor_mp4_mux_file: NAL unit too large"; break; }
      put_nal(nal);
      offsets.push_back(sample_at);
      sizes.push_back(sample_bytes);
      if (type == 5) sync.push_back((uint32_t)sizes.size());
      offset += sample_bytes;
    } else if (type == 6 || type == 9) {
      const uint8_t len[4] = {(uint8_t)(nal.size() >> 24), (uint8_t)(nal.size() >> 16), (uint8_t)(nal.size() >> 8), (uint8_t)nal.size()};
      pending.insert(pending.end(), len, len + 4);
      pending.insert(pending.end(), nal.begin(), nal.end());
    }  // anything else (end of sequence / stream, filler) carries no picture data
  }
  fclose(in);
  if (err.empty() && (sps.size() < 4 || pps.empty())) err = "tor_mp4_mux_file: no SPS/PPS in the stream";
  if (err.empty() && sizes.empty()) err = "tor_mp4_mux_file: no picture in the stream";
  if (!err.empty()) {
    fclose(out);
    remove(dst_mp4_path);
    return fail(err);
  }
  const uint64_t mdat_size = offset - mdat_at;

  // ---- moov ----
  const uint32_t n = (uint32_t)sizes.size();
  const uint32_t timescale = 90000, delta = timescale / (uint32_t)fps;  // mp4.nim:90: 90000 div fps
  const uint64_t duration = (uint64_t)n * delta;
  Box m;
  const size_t moov = m.open("moov");
  {
    const size_t mvhd = m.open_full("mvhd", 1, 0);
    m.u64(0); m.u64(0); m.u32(timescale); m.u64(duration);
    m.u32(0x00010000); m.u16(0x0100); m.u16(0); m.u32(0); m.u32(0);
    unity_matrix(m);
    m.zeros(24);
    m.u32(2);  // next_track_ID
    m.close(mvhd);
    const size_t trak = m.open("trak");
    {
      const size_t tkhd = m.open_full("tkhd", 1, 7);  // enabled | in movie | in preview
      m.u64(0); m.u64(0); m.u32(1); m.u32(0); m.u64(duration);
      m.u32(0); m.u32(0); m.u16(0); m.u16(0); m.u16(0); m.u16(0);
      unity_matrix(m);
      m.u32((uint32_t)width << 16); m.u32((uint32_t)height << 16);
      m.close(tkhd);
      const size_t mdia = m.open("mdia");
      {
        const size_t mdhd = m.open_full("mdhd", 1, 0);
        m.u64(0); m.u64(0); m.u32(timescale); m.u64(duration);
        m.u16(0x55c4); m.u16(0);  // language "und"
        m.close(mdhd);
        const size_t hdlr = m.open_full("hdlr", 0, 0);
        m.u32(0); m.tag("vide"); m.zeros(12);
        const char name[] = "VideoHandler";
        m.bytes((const uint8_t*)name, sizeof name);
        m.close(hdlr);
        const size_t minf = m.open("minf");
        {
          const size_t vmhd = m.open_full("vmhd", 0, 1);
          m.zeros(8);
          m.close(vmhd);
          const size_t dinf = m.open("dinf");
          const size_t dref = m.open_full("dref", 0, 0);
          m.u32(1);
          const size_t url = m.open_full("url ", 0, 1);  // self-contained
          m.close(url);
          m.close(dref);
          m.close(dinf);
          const size_t stbl = m.open("stbl");
          {
            const size_t stsd = m.open_full("stsd", 0, 0);
            m.u32(1);
            const size_t avc1 = m.open("avc1");
            m.zeros(6); m.u16(1);          // data_reference_index
            m.zeros(16);
            m.u16((uint32_t)width); m.u16((uint32_t)height);
            m.u32(0x00480000); m.u32(0x00480000);  // 72 dpi
            m.u32(0); m.u16(1);            // one frame per sample
            m.zeros(32);                   // compressor name
            m.u16(0x0018); m.u16(0xffff);  // depth, pre_defined = -1
            const size_t avcc = m.open("avcC");
            m.u8(1); m.u8(sps[1]); m.u8(sps[2]); m.u8(sps[3]);
            m.u8(0xff);                    // 4-byte NAL lengths
            m.u8(0xe1); m.u16((uint32_t)sps.size()); m.bytes(sps.data(), sps.size());
            m.u8(1); m.u16((uint32_t)pps.size()); m.bytes(pps.data(), pps.size());
            m.close(avcc);
            m.close(avc1);
            m.close(stsd);
            const size_t stts = m.open_full("stts", 0, 0);
            m.u32(1); m.u32(n); m.u32(delta);
            m.close(stts);
            if (sync.size() != n) {  // absent box == every sample is a sync sample
              const size_t stss = m.open_full("stss", 0, 0);
              m.u32((uint32_t)sync.size());
              for (uint32_t s : sync) m.u32(s);
              m.close(stss);
            }
            const size_t stsc = m.open_full("stsc", 0, 0);  // one sample per chunk
            m.u32(1); m.u32(1); m.u32(1); m.u32(1);
            m.close(stsc);
            const size_t stsz = m.open_full("stsz", 0, 0);
            bool same = true;
            for (uint32_t s : sizes) same = same && s == sizes[0];
            if (same) {
              m.u32(sizes[0]); m.u32(n);
            } else {
              m.u32(0); m.u32(n);
              for (uint32_t s : sizes) m.u32(s);
            }
            m.close(stsz);
            const size_t co64 = m.open_full("co64", 0, 0);
            m.u32(n);
            for (uint64_t o : offsets) m.u64(o);
            m.close(co64);
          }
          m.close(stbl);
        }
        m.close(minf);
      }
      m.close(mdia);
    }
    m.close(trak);
  }
  m.close(moov);
  put(m.b.data(), m.b.size());
  // patch the mdat largesize
  if (io_ok && fseek(out, (long)mdat_at + 8, SEEK_SET) == 0) {
    uint8_t be[8];
    for (int i = 0; i < 8; ++i) be[i] = (uint8_t)(mdat_size >> (56 - 8 * i));
    put(be, 8);
  } else {
    io_ok = false;
  }
  if (fclose(out) != 0) io_ok = false;
  if (!io_ok) {
    remove(dst_mp4_path);
    tor::set_last_error(std::string("tor_mp4_mux_file: write error on ") + dst_mp4_path);
    return TOR_ERR_HIP;  // I/O failure: not an argument error
  }
  return (int)n;
}
