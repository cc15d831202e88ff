// fy_seqfile.cpp -- reader / writer for the Hadoop files on either side of the RM2 job, so the
// library can be driven from the reference's own files without a JVM.  Host-only code (no GPU work).
//
// What the reference reads and writes at this seam (M/ = src/main/java/es/udc/fi/dc/irlab/):
//   ratings          SequenceFile<IntPairWritable(user, item), FloatWritable>   M/util/DataInitialization.java:155-183
//   clustering       SequenceFile<IntWritable user, IntWritable cluster>        M/util/DataInitialization.java:199-222, read by
//   clusteringCount  SequenceFile<IntWritable cluster, IntWritable size>        M/common/AbstractByClusterMapper.java:46-75
//   rm2/userSum      SequenceFile<IntWritable, DoubleWritable>                  M/rm/RM2Job.java:110-151
//   rm2/itemColl     MapFile<IntWritable, DoubleWritable> (data + index)        M/rm/RM2Job.java:164-205, M/util/MapFileOutputFormat.java
//   recommendations  SequenceFile<IntPairWritable(user, item), FloatWritable>   M/rm/RM2HDFSReducer.java:44-50
//
// The container is Hadoop 1.2.1's SequenceFile (org.apache.hadoop:hadoop-core:1.2.1, pom.xml:30-34 -- third-party, not in
// /root/reference), version 6, uncompressed, record-oriented, restated from its published format:
//   header   "SEQ" 0x06 | Text key class | Text value class | bool compressed | bool blockCompressed | int32 #metadata
//            (Text pairs) | 16-byte sync marker            (Text = vint length + UTF-8; all integers big-endian)
//   record   int32 recordLength (key + value bytes) | int32 keyLength | key bytes | value bytes
//   sync     int32 -1 | 16-byte marker, written in front of a record once >= 2000 bytes followed the previous one
// IntWritable / FloatWritable / DoubleWritable / LongWritable are the big-endian bytes of the value.
// PARITY UNPINNED at the byte level: the reference holds no binary fixture of any of these files (its tests write them
// with the Hadoop classes at run time).  Mahout 0.8's IntPairWritable (org.apache.mahout:mahout-core:0.8, pom.xml:15-19, absent too)
// is restated as two big-endian int32 (first, second), which is what its write(DataOutput) emits to the best of my
// knowledge; a file written by a real Mahout must be checked against this before it is trusted.
#include <dirent.h>
#include <sys/stat.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../../include/filmyou.h"

namespace fy {
void set_error(const char* fmt, ...);
}

namespace {

const char* K_INT = "org.apache.hadoop.io.IntWritable";
const char* K_FLOAT = "org.apache.hadoop.io.FloatWritable";
const char* K_DOUBLE = "org.apache.hadoop.io.DoubleWritable";
const char* K_LONG = "org.apache.hadoop.io.LongWritable";
const char* K_INTPAIR = "org.apache.mahout.common.IntPairWritable";
constexpr int SYNC_INTERVAL = 2000;   // SequenceFile.SYNC_INTERVAL = 100 * (4 + 16)
constexpr int INDEX_INTERVAL = 128;   // MapFile "io.map.index.interval" default

struct Fail {
    std::string msg;
};

void put_be32(std::vector<uint8_t>& b, uint32_t v) { for (int s = 24; s >= 0; s -= 8) b.push_back((uint8_t)(v >> s)); }
void put_be64(std::vector<uint8_t>& b, uint64_t v) { for (int s = 56; s >= 0; s -= 8) b.push_back((uint8_t)(v >> s)); }
uint32_t get_be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
uint64_t get_be64(const uint8_t* p) { return ((uint64_t)get_be32(p) << 32) | get_be32(p + 4); }

// WritableUtils.writeVLong (the length prefix of Text)
void put_vlong(std::vector<uint8_t>& b, int64_t i) {
    if (i >= -112 && i <= 127) { b.push_back((uint8_t)i); return; }
    int len = -112;
    if (i < 0) { i ^= -1LL; len = -120; }
    int64_t tmp = i;
    while (tmp != 0) { tmp >>= 8; len--; }
    b.push_back((uint8_t)len);
    len = (len < -120) ? -(len + 120) : -(len + 112);
    for (int idx = len; idx != 0; idx--) b.push_back((uint8_t)((i >> ((idx - 1) * 8)) & 0xFF));
}
int64_t get_vlong(const uint8_t*& p, const uint8_t* end) {
    if (p >= end) throw Fail{"truncated vint"};
    const int8_t first = (int8_t)*p++;
    if (first >= -112) return first;
    const bool neg = first < -120;
    const int len = neg ? -(first + 120) : -(first + 112);
    if (p + len > end) throw Fail{"truncated vint"};
    int64_t v = 0;
    for (int k = 0; k < len; k++) v = (v << 8) | *p++;
    return neg ? (v ^ -1LL) : v;
}
void put_text(std::vector<uint8_t>& b, const char* s) {
    const size_t n = std::strlen(s);
    put_vlong(b, (int64_t)n);
    b.insert(b.end(), s, s + n);
}
std::string get_text(const uint8_t*& p, const uint8_t* end) {
    const int64_t n = get_vlong(p, end);
    if (n < 0 || p + n > end) throw Fail{"truncated Text"};
    std::string s(reinterpret_cast<const char*>(p), (size_t)n);
    p += n;
    return s;
}

// ---------------------------------------------------------------- writer
struct Writer {
    FILE* f = nullptr;
    uint8_t sync[16];
    int64_t pos = 0, last_sync = 0;
    std::string path;

    Writer(const std::string& p, const char* key_class, const char* value_class) : path(p) {
        f = std::fopen(p.c_str(), "wb");
        if (!f) throw Fail{"cannot create " + p};
        std::random_device rd;     // Hadoop hashes a UID and the time; any 16 bytes that do not occur in the data will do
        for (auto& x : sync) x = (uint8_t)rd();
        std::vector<uint8_t> h = {'S', 'E', 'Q', 6};
        put_text(h, key_class);
        put_text(h, value_class);
        h.push_back(0);            // compression
        h.push_back(0);            // block compression
        put_be32(h, 0);            // metadata entries
        h.insert(h.end(), sync, sync + 16);
        write(h);
        last_sync = pos;           // the header ends with a sync marker
    }
    ~Writer() { if (f) std::fclose(f); }
    void write(const std::vector<uint8_t>& b) {
        if (!b.empty() && std::fwrite(b.data(), 1, b.size(), f) != b.size()) throw Fail{"write failed: " + path};
        pos += (int64_t)b.size();
    }
    // returns the position of the record (in front of its sync escape, if one was due): what MapFile's index stores
    int64_t append(const uint8_t* key, int klen, const uint8_t* val, int vlen) {
        const int64_t at = pos;
        std::vector<uint8_t> r;
        if (pos >= last_sync + SYNC_INTERVAL) {      // SequenceFile.Writer.checkAndWriteSync
            put_be32(r, 0xFFFFFFFFu);
            r.insert(r.end(), sync, sync + 16);
            last_sync = pos + 20;
        }
        put_be32(r, (uint32_t)(klen + vlen));
        put_be32(r, (uint32_t)klen);
        r.insert(r.end(), key, key + klen);
        r.insert(r.end(), val, val + vlen);
        write(r);
        return at;
    }
    void close() {
        if (f && std::fclose(f) != 0) { f = nullptr; throw Fail{"close failed: " + path}; }
        f = nullptr;
    }
};

// ---------------------------------------------------------------- reader
struct Records {
    std::string key_class, value_class;
    std::vector<uint8_t> keys, values;     // fixed-size records only (every type at this seam is fixed-size)
    int klen = -1, vlen = -1;
    int64_t n = 0;
};

void read_one_file(const std::string& path, Records& R) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) throw Fail{"cannot open " + path};
    std::vector<uint8_t> buf;
    {
        std::fseek(f, 0, SEEK_END);
        const long sz = std::ftell(f);
        std::fseek(f, 0, SEEK_SET);
        buf.resize((size_t)std::max<long>(0, sz));
        const size_t got = buf.empty() ? 0 : std::fread(buf.data(), 1, buf.size(), f);
        std::fclose(f);
        if (got != buf.size()) throw Fail{"short read: " + path};
    }
    const uint8_t* p = buf.data();
    const uint8_t* end = p + buf.size();
    if (buf.size() < 4 || p[0] != 'S' || p[1] != 'E' || p[2] != 'Q') throw Fail{path + " is not a SequenceFile"};
    const int version = p[3];
    if (version != 6 && version != 5) throw Fail{path + ": SequenceFile version " + std::to_string(version) + " is not supported (expected 6)"};
    p += 4;
    const std::string kc = get_text(p, end), vc = get_text(p, end);
    if (p + 2 > end) throw Fail{"truncated header: " + path};
    const bool compressed = *p++ != 0, block = *p++ != 0;
    if (compressed || block) throw Fail{path + " is compressed: only the uncompressed record format the reference's jobs write is supported"};
    if (version >= 6) {
        if (p + 4 > end) throw Fail{"truncated header: " + path};
        const uint32_t nmeta = get_be32(p);
        p += 4;
        for (uint32_t k = 0; k < nmeta; k++) { (void)get_text(p, end); (void)get_text(p, end); }
    }
    if (p + 16 > end) throw Fail{"truncated header: " + path};
    uint8_t sync[16];
    std::memcpy(sync, p, 16);
    p += 16;
    if (R.key_class.empty()) { R.key_class = kc; R.value_class = vc; }
    else if (R.key_class != kc || R.value_class != vc) throw Fail{path + ": key/value classes differ from the other part files"};
    while (p < end) {
        if (p + 4 > end) throw Fail{"truncated record: " + path};
        const uint32_t rl = get_be32(p);
        p += 4;
        if (rl == 0xFFFFFFFFu) {      // sync escape
            if (p + 16 > end || std::memcmp(p, sync, 16) != 0) throw Fail{"corrupt sync marker: " + path};
            p += 16;
            continue;
        }
        if (p + 4 > end) throw Fail{"truncated record: " + path};
        const uint32_t kl = get_be32(p);
        p += 4;
        if (kl > rl || p + rl > end) throw Fail{"corrupt record length: " + path};
        const int vl = (int)(rl - kl);
        if (R.klen < 0) { R.klen = (int)kl; R.vlen = vl; }
        if ((int)kl != R.klen || vl != R.vlen) throw Fail{path + ": variable-size records are not supported"};
        R.keys.insert(R.keys.end(), p, p + kl);
        R.values.insert(R.values.end(), p + kl, p + rl);
        p += rl;
        R.n++;
    }
}

// A path is a file, a MapFile directory (its "data" file is read) or a job output directory (every part file, in name
// order; names starting with '_' or '.' -- _SUCCESS, _logs, .crc -- are skipped, like FileInputFormat's hidden-file filter).
void read_path(const std::string& path, Records& R) {
    struct stat st;
    if (stat(path.c_str(), &st) != 0) throw Fail{"no such file or directory: " + path};
    if (!S_ISDIR(st.st_mode)) { read_one_file(path, R); return; }
    std::vector<std::string> names;
    DIR* d = opendir(path.c_str());
    if (!d) throw Fail{"cannot list " + path};
    bool has_data = false, has_index = false;
    while (dirent* e = readdir(d)) {
        const std::string n = e->d_name;
        if (n.empty() || n[0] == '.' || n[0] == '_') continue;
        if (n == "data") has_data = true;
        if (n == "index") has_index = true;
        names.push_back(n);
    }
    closedir(d);
    std::sort(names.begin(), names.end());
    if (has_data && has_index) { read_one_file(path + "/data", R); return; }     // a MapFile
    for (const auto& n : names) {
        const std::string p = path + "/" + n;
        if (stat(p.c_str(), &st) != 0) continue;
        if (S_ISDIR(st.st_mode)) read_path(p, R);      // part-r-00000 of a MapFileOutputFormat job is itself a MapFile directory
        else read_one_file(p, R);
    }
}

void mkdirs(const std::string& path) {
    std::string cur;
    for (size_t i = 0; i <= path.size(); i++) {
        if (i == path.size() || path[i] == '/') {
            if (!cur.empty() && cur != "/") (void)mkdir(cur.c_str(), 0777);
        }
        if (i < path.size()) cur.push_back(path[i]);
    }
}

void expect(const Records& R, const char* kc, const char* vc, int kl, int vl, const char* path) {
    if (R.n == 0) return;
    if (R.key_class != kc || R.value_class != vc)
        throw Fail{std::string(path) + " holds <" + R.key_class + ", " + R.value_class + ">, expected <" + kc + ", " + vc + ">"};
    if (R.klen != kl || R.vlen != vl) throw Fail{std::string(path) + ": unexpected record size"};
}

template <class T>
T* dup(const std::vector<T>& v) {
    T* p = static_cast<T*>(std::malloc(std::max<size_t>(1, v.size()) * sizeof(T)));
    if (!p) throw Fail{"out of host memory"};
    if (!v.empty()) std::memcpy(p, v.data(), v.size() * sizeof(T));
    return p;
}

}  // namespace

#define FY_SEQ_TRY try {
#define FY_SEQ_CATCH                                          \
    return FY_OK;                                             \
    }                                                         \
    catch (const Fail& e) {                                   \
        fy::set_error("%s", e.msg.c_str());                   \
        return FY_ERR_IO;                                     \
    }                                                         \
    catch (const std::exception& e) {                         \
        fy::set_error("%s", e.what());                        \
        return FY_ERR_IO;                                     \
    }

extern "C" {

void fy_buffer_free(void* p) { std::free(p); }

int fy_seqfile_read_int_int(const char* path, int64_t* n, int32_t** key, int32_t** value) {
    if (!path || !n || !key || !value) { fy::set_error("NULL argument"); return FY_ERR_INVALID_ARGUMENT; }
    FY_SEQ_TRY
    Records R;
    read_path(path, R);
    expect(R, K_INT, K_INT, 4, 4, path);
    std::vector<int32_t> k((size_t)R.n), v((size_t)R.n);
    for (int64_t i = 0; i < R.n; i++) { k[i] = (int32_t)get_be32(&R.keys[4 * i]); v[i] = (int32_t)get_be32(&R.values[4 * i]); }
    *n = R.n;
    *key = dup(k);
    *value = dup(v);
    FY_SEQ_CATCH
}

int fy_seqfile_read_int_double(const char* path, int64_t* n, int32_t** key, double** value) {
    if (!path || !n || !key || !value) { fy::set_error("NULL argument"); return FY_ERR_INVALID_ARGUMENT; }
    FY_SEQ_TRY
    Records R;
    read_path(path, R);
    expect(R, K_INT, K_DOUBLE, 4, 8, path);
    std::vector<int32_t> k((size_t)R.n);
    std::vector<double> v((size_t)R.n);
    for (int64_t i = 0; i < R.n; i++) {
        k[i] = (int32_t)get_be32(&R.keys[4 * i]);
        const uint64_t bits = get_be64(&R.values[8 * i]);
        std::memcpy(&v[i], &bits, 8);
    }
    *n = R.n;
    *key = dup(k);
    *value = dup(v);
    FY_SEQ_CATCH
}

int fy_seqfile_read_intpair_float(const char* path, int64_t* n, int32_t** first, int32_t** second, float** value) {
    if (!path || !n || !first || !second || !value) { fy::set_error("NULL argument"); return FY_ERR_INVALID_ARGUMENT; }
    FY_SEQ_TRY
    Records R;
    read_path(path, R);
    expect(R, K_INTPAIR, K_FLOAT, 8, 4, path);
    std::vector<int32_t> a((size_t)R.n), b((size_t)R.n);
    std::vector<float> v((size_t)R.n);
    for (int64_t i = 0; i < R.n; i++) {
        a[i] = (int32_t)get_be32(&R.keys[8 * i]);
        b[i] = (int32_t)get_be32(&R.keys[8 * i + 4]);
        const uint32_t bits = get_be32(&R.values[4 * i]);
        std::memcpy(&v[i], &bits, 4);
    }
    *n = R.n;
    *first = dup(a);
    *second = dup(b);
    *value = dup(v);
    FY_SEQ_CATCH
}

int fy_seqfile_write_int_int(const char* file, int64_t n, const int32_t* key, const int32_t* value) {
    if (!file || n < 0 || (n && (!key || !value))) { fy::set_error("bad argument"); return FY_ERR_INVALID_ARGUMENT; }
    FY_SEQ_TRY
    const std::string p = file;
    if (p.find('/') != std::string::npos) mkdirs(p.substr(0, p.rfind('/')));
    Writer W(p, K_INT, K_INT);
    for (int64_t i = 0; i < n; i++) {
        std::vector<uint8_t> k, v;
        put_be32(k, (uint32_t)key[i]);
        put_be32(v, (uint32_t)value[i]);
        W.append(k.data(), 4, v.data(), 4);
    }
    W.close();
    FY_SEQ_CATCH
}

int fy_seqfile_write_int_double(const char* file, int64_t n, const int32_t* key, const double* value) {
    if (!file || n < 0 || (n && (!key || !value))) { fy::set_error("bad argument"); return FY_ERR_INVALID_ARGUMENT; }
    FY_SEQ_TRY
    const std::string p = file;
    if (p.find('/') != std::string::npos) mkdirs(p.substr(0, p.rfind('/')));
    Writer W(p, K_INT, K_DOUBLE);
    for (int64_t i = 0; i < n; i++) {
        std::vector<uint8_t> k, v;
        uint64_t bits;
        std::memcpy(&bits, &value[i], 8);
        put_be32(k, (uint32_t)key[i]);
        put_be64(v, bits);
        W.append(k.data(), 4, v.data(), 8);
    }
    W.close();
    FY_SEQ_CATCH
}

int fy_seqfile_write_intpair_float(const char* file, int64_t n, const int32_t* first, const int32_t* second, const float* value) {
    if (!file || n < 0 || (n && (!first || !second || !value))) { fy::set_error("bad argument"); return FY_ERR_INVALID_ARGUMENT; }
    FY_SEQ_TRY
    const std::string p = file;
    if (p.find('/') != std::string::npos) mkdirs(p.substr(0, p.rfind('/')));
    Writer W(p, K_INTPAIR, K_FLOAT);
    for (int64_t i = 0; i < n; i++) {
        std::vector<uint8_t> k, v;
        uint32_t bits;
        std::memcpy(&bits, &value[i], 4);
        put_be32(k, (uint32_t)first[i]);
        put_be32(k, (uint32_t)second[i]);
        put_be32(v, bits);
        W.append(k.data(), 8, v.data(), 4);
    }
    W.close();
    FY_SEQ_CATCH
}

/* MapFile<IntWritable, DoubleWritable>: directory with `data` (the records, ascending keys -- MapFile.Writer refuses
 * anything else) and `index` (every 128th key with the position of its record in `data`, LongWritable). */
int fy_mapfile_write_int_double(const char* dir, int64_t n, const int32_t* key, const double* value) {
    if (!dir || n < 0 || (n && (!key || !value))) { fy::set_error("bad argument"); return FY_ERR_INVALID_ARGUMENT; }
    FY_SEQ_TRY
    for (int64_t i = 1; i < n; i++)
        if (key[i] <= key[i - 1]) throw Fail{"MapFile keys must be strictly ascending (key out of order: " + std::to_string(key[i]) + ")"};
    mkdirs(dir);
    Writer D(std::string(dir) + "/data", K_INT, K_DOUBLE), I(std::string(dir) + "/index", K_INT, K_LONG);
    for (int64_t i = 0; i < n; i++) {
        std::vector<uint8_t> k, v;
        uint64_t bits;
        std::memcpy(&bits, &value[i], 8);
        put_be32(k, (uint32_t)key[i]);
        put_be64(v, bits);
        const int64_t at = D.append(k.data(), 4, v.data(), 8);
        if (i % INDEX_INTERVAL == 0) {
            std::vector<uint8_t> pos;
            put_be64(pos, (uint64_t)at);
            I.append(k.data(), 4, pos.data(), 8);
        }
    }
    D.close();
    I.close();
    FY_SEQ_CATCH
}

}  // extern "C"
