// inspect.cpp — `sylph-hip inspect`: YAML summary of *.syldb / *.sylsp files (src/inspect.rs:1-233).  Host only: no GPU is
// touched.  The reference serialises two vectors of structs with serde_yaml 0.9.34 (a third-party crate that is not under
// /root/reference: its scalar styles and ryu's float formatting are restated from their documented behaviour — parity with a
// reference binary is unpinned; the reference's own test only looks for the file names in the output,
// tests/integration_test.rs:505-549).  Field order = declaration order of SequencesSketchInspect / DatabaseSketch /
// GenomeSketchInspect (inspect.rs:19-77).
#include <charconv>
#include <cmath>
#include <cstring>

#include "sylph_host.hpp"

namespace sylph_host {

namespace {

void warn(const std::string& m) { fprintf(stderr, "WARN  [sylph_hip] %s\n", m.c_str()); }
void info(const std::string& m) { fprintf(stderr, "INFO  [sylph_hip] %s\n", m.c_str()); }

bool ends_with(const std::string& s, const char* suf) {
    const size_t n = strlen(suf);
    return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

// shortest round-trip decimal digits of v (> 0, finite): digits d1..dn and the power of ten of the LAST digit
template <class T>
void shortest_digits(T v, std::string& digits, int& k) {
    char buf[64];
    auto r = std::to_chars(buf, buf + sizeof buf, v, std::chars_format::scientific);
    std::string s(buf, r.ptr);                      // d[.ddd]e±XX
    const size_t e = s.find('e');
    std::string mant = s.substr(0, e);
    const int exp10 = atoi(s.c_str() + e + 1);
    digits.clear();
    for (char c : mant) if (c != '.') digits.push_back(c);
    k = exp10 - ((int)digits.size() - 1);
}

// ryu's "pretty" formatting (what serde_yaml hands to the emitter for f32 / f64): fixed notation while the decimal point
// stays within `hi` digits (16 for f64, 13 for f32) and the leading zeros within `lo` (-5 / -6), exponent form otherwise
template <class T>
std::string ryu_pretty(T v, int hi, int lo) {
    if (std::isnan(v)) return ".nan";
    if (std::isinf(v)) return v > 0 ? ".inf" : "-.inf";
    if (v == 0) return std::signbit(v) ? "-0.0" : "0.0";
    std::string out = v < 0 ? "-" : "";
    std::string d;
    int k;
    shortest_digits(v < 0 ? -v : v, d, k);
    const int len = (int)d.size(), kk = len + k;     // kk: position of the decimal point relative to the first digit
    if (0 <= k && kk <= hi) return out + d + std::string((size_t)k, '0') + ".0";
    if (0 < kk && kk <= hi) return out + d.substr(0, (size_t)kk) + "." + d.substr((size_t)kk);
    if (lo < kk && kk <= 0) return out + "0." + std::string((size_t)(-kk), '0') + d;
    if (len == 1) return out + d + "e" + std::to_string(kk - 1);
    return out + d.substr(0, 1) + "." + d.substr(1) + "e" + std::to_string(kk - 1);
}

// would an unquoted scalar read back as null / bool / number (YAML 1.2 core schema as serde_yaml resolves it)?
bool resolves_to_non_string(const std::string& s) {
    static const char* words[] = {"null", "Null", "NULL", "~", "true", "True", "TRUE", "false", "False", "FALSE",
                                  ".inf", ".Inf", ".INF", "+.inf", "+.Inf", "+.INF", "-.inf", "-.Inf", "-.INF", ".nan", ".NaN", ".NAN"};
    for (const char* w : words) if (s == w) return true;
    if (s.empty()) return false;
    size_t i = (s[0] == '+' || s[0] == '-') ? 1 : 0;
    if (i >= s.size()) return false;
    if (s.compare(i, 2, "0x") == 0 || s.compare(i, 2, "0o") == 0 || s.compare(i, 2, "0b") == 0) return s.size() > i + 2;
    bool digit = false, dot = false, exp = false;
    for (; i < s.size(); i++) {
        const char c = s[i];
        if (c >= '0' && c <= '9') digit = true;
        else if (c == '.' && !dot && !exp) dot = true;
        else if ((c == 'e' || c == 'E') && digit && !exp) { exp = true; digit = false; if (i + 1 < s.size() && (s[i + 1] == '+' || s[i + 1] == '-')) i++; }
        else if (c == '_') continue;                // digits_but_not_number: quoted to stay a string
        else return false;
    }
    return digit;
}

// block-context scalar the way libyaml's emitter chooses it: plain when that is unambiguous, else single-quoted, double-quoted
// only for characters single quotes cannot carry
std::string yaml_str(const std::string& s) {
    bool special = false, plain_ok = !s.empty();
    for (unsigned char c : s) if (c < 0x20 || c == 0x7F) special = true;
    if (special) {
        std::string o = "\"";
        for (unsigned char c : s) {
            if (c == '"') o += "\\\"";
            else if (c == '\\') o += "\\\\";
            else if (c == '\n') o += "\\n";
            else if (c == '\t') o += "\\t";
            else if (c == '\r') o += "\\r";
            else if (c < 0x20 || c == 0x7F) { char b[8]; snprintf(b, sizeof b, "\\x%02X", c); o += b; }
            else o.push_back((char)c);
        }
        return o + "\"";
    }
    if (plain_ok) {
        const char f = s.front();
        if (s.front() == ' ' || s.back() == ' ') plain_ok = false;
        else if (strchr(",[]{}#&*!|>'\"%@`", f)) plain_ok = false;
        else if ((f == '-' || f == '?' || f == ':') && (s.size() == 1 || s[1] == ' ')) plain_ok = false;
        else if (s.find(": ") != std::string::npos || s.find(" #") != std::string::npos || s.back() == ':') plain_ok = false;
        else if (s == "---" || s == "...") plain_ok = false;
        else if (resolves_to_non_string(s)) plain_ok = false;
    }
    if (plain_ok) return s;
    std::string o = "'";
    for (char c : s) { if (c == '\'') o += "''"; else o.push_back(c); }
    return o + "'";
}

}  // namespace

std::string inspect_f32(float v) { return ryu_pretty<float>(v, 13, -6); }
std::string inspect_f64(double v) { return ryu_pretty<double>(v, 16, -5); }
std::string inspect_str(const std::string& s) { return yaml_str(s); }

// inspect.rs:117-177
int inspect(const InspectArgs& args, FILE* out) {
    std::vector<std::string> read_sketch_files, genome_sketch_files;
    for (const auto& f : args.files) {
        if (ends_with(f, ".syldb") || ends_with(f, ".sylqueries")) genome_sketch_files.push_back(f);
        else if (ends_with(f, ".sylsp") || ends_with(f, ".sylsample")) read_sketch_files.push_back(f);
        else warn(f + " file is not a .sylsp or .syldb file. Skipping...");
    }
    std::string yaml;
    for (const auto& f : genome_sketch_files) {                              // get_db_sketch_inspect, :179-212
        const std::vector<GenomeSketch> gs = read_syldb(f);
        if (gs.empty()) {                                                    // DatabaseSketch::default()
            warn("The database sketch `" + f + "` is empty. Skipping...");
            yaml += "- database_file: ''\n  c: 0\n  k: 0\n  min_spacing_parameter: 0\n  genome_files: []\n";
            continue;
        }
        info("Database file " + f + " processed with " + std::to_string(gs.size()) + " genomes");
        yaml += "- database_file: " + yaml_str(f) + "\n";
        yaml += "  c: " + std::to_string(gs.front().c) + "\n  k: " + std::to_string(gs.front().k) + "\n";
        yaml += "  min_spacing_parameter: " + std::to_string(gs.front().min_spacing) + "\n  genome_files:\n";
        for (const auto& g : gs) {
            yaml += "  - file_name: " + yaml_str(g.file_name) + "\n";
            yaml += "    genome_kmers_num: " + std::to_string(g.genome_kmers.size()) + "\n";
            yaml += "    first_contig_name: " + yaml_str(g.first_contig_name) + "\n";
            yaml += "    genome_size: " + std::to_string(g.gn_size) + "\n";
        }
    }
    for (const auto& f : read_sketch_files) {                                // get_seq_sketch_inspect, :214-233; From<SequencesSketch>, :30-46
        const SequencesSketch s = read_sylsp(f);
        info("Sequence file " + f + " processed");
        const float approx = (float)(s.mean_read_length + (double)s.k - 1.) / (float)s.mean_read_length * (float)s.c * (float)s.kmers.size();
        yaml += "- file_name: " + yaml_str(s.file_name) + "\n";
        yaml += "  c: " + std::to_string(s.c) + "\n  k: " + std::to_string(s.k) + "\n";
        yaml += "  num_sketched_kmers: " + std::to_string(s.kmers.size()) + "\n";
        yaml += "  approximate_number_bases: " + inspect_f32(approx) + "\n";
        yaml += "  mean_read_length: " + inspect_f64(s.mean_read_length) + "\n";
        yaml += "  sample_name: " + (s.sample_name ? yaml_str(*s.sample_name) : std::string("null")) + "\n";
        yaml += std::string("  paired: ") + (s.paired ? "true" : "false") + "\n";
    }
    if (fwrite(yaml.data(), 1, yaml.size(), out) != yaml.size()) throw Error{1, "could not write the inspect output"};
    fflush(out);
    return 0;
}

}  // namespace sylph_host
