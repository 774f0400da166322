// main.cpp — `sylph-hip sketch|profile|query`: the reference's command surface for the hot path only (flag names and
// defaults from cmdline.rs:28-173; hidden estimators and logging flags are not carried).
#include <cstdlib>
#include <unistd.h>
#include <fcntl.h>
#include <sys/wait.h>
#include <cerrno>
#include <cstring>
#include <functional>

#include "sylph_host.hpp"

using namespace sylph_host;

namespace {

struct Argv {
    int argc; char** argv; int i = 2;
    bool more() const { return i < argc; }
    std::string cur() const { return argv[i]; }
    // values of a `multiple=true` option: everything up to the next token starting with '-'
    std::vector<std::string> multi() {
        std::vector<std::string> v;
        i++;
        while (i < argc && !(argv[i][0] == '-' && strlen(argv[i]) > 1)) v.push_back(argv[i++]);
        return v;
    }
    std::string one() {
        if (i + 1 >= argc) throw Error{2, std::string("option ") + argv[i] + " needs a value"};
        i += 2;
        return argv[i - 1];
    }
};

void append(std::vector<std::string>& dst, const std::vector<std::string>& src) { dst.insert(dst.end(), src.begin(), src.end()); }

int run_sketch(Argv a) {
    SketchArgs s;
    while (a.more()) {
        const std::string t = a.cur();
        if (t == "-o" || t == "--out-name-db") s.db_out_name = a.one();
        else if (t == "-d" || t == "--sample-output-directory") s.sample_output_dir = a.one();
        else if (t == "-i" || t == "--individual-records") { s.individual = true; a.i++; }
        else if (t == "-r" || t == "--reads") append(s.reads, a.multi());
        else if (t == "-g" || t == "--genomes") append(s.genomes, a.multi());
        else if (t == "-l" || t == "--list") s.list_sequence = a.one();
        else if (t == "--rl") s.list_reads = a.one();
        else if (t == "--gl") s.list_genomes = a.one();
        else if (t == "--l1") s.list_first_pair = a.one();
        else if (t == "--l2") s.list_second_pair = a.one();
        else if (t == "--lS") s.list_sample_names = a.one();
        else if (t == "-S" || t == "--sample-names") { if (!s.sample_names) s.sample_names.emplace(); append(*s.sample_names, a.multi()); }
        else if (t == "-k") s.k = strtoull(a.one().c_str(), nullptr, 10);
        else if (t == "-c") s.c = strtoull(a.one().c_str(), nullptr, 10);
        else if (t == "-t") s.threads = std::max<uint64_t>(1, strtoull(a.one().c_str(), nullptr, 10));   // samples in flight
        else if (t == "--no-dedup") { s.no_dedup = true; a.i++; }
        else if (t == "--disable-profiling") { s.no_pseudotax = true; a.i++; }
        else if (t == "--min-spacing") s.min_spacing_kmer = strtoull(a.one().c_str(), nullptr, 10);
        else if (t == "--fpr") s.fpr = atof(a.one().c_str());
        else if (t == "--exact-dedup") { s.exact_dedup = true; a.i++; }
        else if (t == "--gpus") { const std::string v = a.one(); s.gpus = v == "all" ? -1 : std::max(1, atoi(v.c_str())); }
        else if (t == "-1" || t == "--first-pairs") append(s.first_pair, a.multi());
        else if (t == "-2" || t == "--second-pairs") append(s.second_pair, a.multi());
        else if (t == "--debug" || t == "--trace") a.i++;
        else if (t[0] == '-' && t.size() > 1) throw Error{2, "unknown option " + t};
        else { s.files.push_back(t); a.i++; }
    }
    // (on the heap, and only destroyed on the long way out: tearing the context down — queues, pools, page-locked buffers — is part of
    //  what main() skips by leaving through _exit)
    Engine* e = new Engine();
    const int rc = sketch(*e, s);
    if (!fast_exit()) delete e;
    return rc;
}

int run_contain(Argv a, bool profile) {
    ContainCmdArgs c;
    while (a.more()) {
        const std::string t = a.cur();
        if (t == "-l" || t == "--list") c.file_list = a.one();
        else if (t == "--min-count-correct") c.min_count_correct = atof(a.one().c_str());
        else if (t == "-M" || t == "--min-number-kmers") c.min_number_kmers = atof(a.one().c_str());
        else if (t == "-m" || t == "--minimum-ani") c.minimum_ani = atof(a.one().c_str());
        else if (t == "-t") c.threads = std::max<uint64_t>(1, strtoull(a.one().c_str(), nullptr, 10));
        else if (t == "-s" || t == "--sample-threads") a.one();
        else if (t == "-u" || t == "--estimate-unknown") { c.estimate_unknown = true; a.i++; }
        else if (t == "-I" || t == "--read-seq-id") c.seq_id = atof(a.one().c_str());
        else if (t == "-R" || t == "--redundancy-threshold") c.redundant_ani = atof(a.one().c_str());
        else if (t == "-r" || t == "--reads") append(c.reads, a.multi());
        else if (t == "-1" || t == "--first-pairs") append(c.first_pair, a.multi());
        else if (t == "-2" || t == "--second-pairs") append(c.second_pair, a.multi());
        else if (t == "-c") c.c = strtoull(a.one().c_str(), nullptr, 10);
        else if (t == "-k") c.k = strtoull(a.one().c_str(), nullptr, 10);
        else if (t == "-i" || t == "--individual-records") { c.individual = true; a.i++; }
        else if (t == "--min-spacing") c.min_spacing_kmer = strtoull(a.one().c_str(), nullptr, 10);
        else if (t == "-o" || t == "--output-file") c.out_file_name = a.one();
        else if (t == "--no-ci") { c.no_ci = true; a.i++; }
        else if (t == "--no-adjust") { c.no_adj = true; a.i++; }
        else if (t == "--mean-coverage") { c.mean_coverage = true; a.i++; }
        else if (t == "--debug-f64") { c.debug_f64 = true; a.i++; }
        else if (t == "--exact-dedup") { c.exact_dedup = true; a.i++; }
        else if (t == "--gpus") { const std::string v = a.one(); c.gpus = v == "all" ? -1 : std::max(1, atoi(v.c_str())); }
        else if (t == "--debug" || t == "--trace" || t == "--log-reassignments") a.i++;
        else if (t[0] == '-' && t.size() > 1) throw Error{2, "unknown option " + t};
        else { c.files.push_back(t); a.i++; }
    }
    FILE* out = stdout;
    if (c.out_file_name) {
        out = fopen(c.out_file_name->c_str(), "w");
        if (!out) throw Error{1, "could not create " + *c.out_file_name};
    }
    // (profile / query give their device memory back themselves, context included: see DbGuard in commands.cpp)
    Engine* e = new Engine();
    const int rc = contain(*e, c, profile, out);
    if (out != stdout) fclose(out);
    delete e;
    return rc;
}

int run_inspect(Argv a) {   // cmdline.rs:166-173; no GPU involved
    InspectArgs c;
    while (a.more()) {
        const std::string t = a.cur();
        if (t == "-o" || t == "--output-file") c.out_file_name = a.one();
        else if (t[0] == '-' && t.size() > 1) throw Error{2, "unknown option " + t};
        else { c.files.push_back(t); a.i++; }
    }
    FILE* out = stdout;
    if (c.out_file_name) {
        out = fopen(c.out_file_name->c_str(), "w");
        if (!out) throw Error{1, "could not create " + *c.out_file_name};
    }
    const int rc = inspect(c, out);
    if (out != stdout) fclose(out);
    return rc;
}

}  // namespace

// The work runs in a CHILD process (forked before anything touches the GPU runtime); the process the user started waits for one word
// from it — the exit status, sent when every output is written and the standard streams are flushed — and leaves at once.  What the
// child still has to do then is give its address space back: GBs of file mappings, inflated copies and index arrays, which the kernel
// frees page by page (0.15-0.2 s after a four-sample or a .gz command, measured in round 5: profiles/r05_cli_first_sample_trace.txt) —
// it does that as an orphan, with its standard streams closed (so that a caller reading our pipes sees end-of-file when WE exit).
// `sylph-hip sketch` only (see run_in_child), and OPT-IN since round 6 (SYLPH_HIP_FORK=1): the process the user started must be the
// one that does the work — a signal sent to it has to stop the work, `time` / getrusage / a scheduler's accounting have to see its
// CPU and memory, and the next command must not find the previous one's HBM and RAM still being torn down by an orphan.
// By default everything runs in one process, whose wall clock includes its own teardown, as the reference's does.
static int g_report_fd = -1;
static int run_in_child(int argc, char** argv) {
    // `sketch` only: what a `profile` / `query` leaves behind is a 29 GB index in HBM and a 14 GB mapping — with those still being torn
    // down by an orphan, the NEXT command's database load took 3-5 s instead of 1.3 (profiles/r05_db_load.txt, first version of this)
    const char* want = getenv("SYLPH_HIP_FORK");
    if (!want || atoi(want) == 0 || !fast_exit() || getenv("SYLPH_HIP_NO_FORK") || argc < 2 || strcmp(argv[1], "sketch") != 0) return -1;
    int fds[2];
    if (pipe(fds) != 0) return -1;
    fflush(stdout);
    fflush(stderr);
    const pid_t pid = fork();
    if (pid < 0) { close(fds[0]); close(fds[1]); return -1; }
    if (pid == 0) { close(fds[0]); g_report_fd = fds[1]; return -1; }        // the child: does the work, reports through g_report_fd
    close(fds[1]);
    int rc = 0;
    ssize_t got;
    do { got = read(fds[0], &rc, sizeof(rc)); } while (got < 0 && errno == EINTR);
    if (got == (ssize_t)sizeof(rc)) _exit(rc);                                // outputs complete: leave, the child cleans up alone
    int st = 0;                                                               // the child died without a word: its status is ours
    while (waitpid(pid, &st, 0) < 0 && errno == EINTR) {}
    _exit(WIFEXITED(st) ? WEXITSTATUS(st) : 128 + (WIFSIGNALED(st) ? WTERMSIG(st) : 0));
}

int main(int argc, char** argv) {
    (void)run_in_child(argc, argv);          // (the parent never returns from here)
    trace_mark("main()");
    if (argc < 2) { fprintf(stderr, "usage: sylph-hip <sketch|profile|query|inspect> ...\n"); return 2; }
    int rc = 2;
    try {
        const std::string cmd = argv[1];
        Argv a{argc, argv};
        if (cmd == "sketch") rc = run_sketch(a);
        else if (cmd == "profile") rc = run_contain(a, true);
        else if (cmd == "query") rc = run_contain(a, false);
        else if (cmd == "inspect") rc = run_inspect(a);
        else fprintf(stderr, "unknown subcommand %s\n", argv[1]);
    } catch (const Error& e) {
        fprintf(stderr, "ERROR [sylph_hip] %s\n", e.msg.c_str());   // log::error! + std::process::exit(1) in the reference
        rc = e.code;
    }
    trace_mark("main: done");
    // Everything the command produces has been written and closed by now.  Leaving through exit() would run the HIP runtime's
    // teardown (queues, VM, page-locked buffers: 0.1-0.15 s of a 0.5 s one-sample command, measured in round 5) for memory the
    // kernel reclaims with the process anyway: flush the standard streams and leave.  SYLPH_HIP_CLEAN_EXIT=1 keeps the long way
    // (leak checkers).
    fflush(stdout);
    fflush(stderr);
    if (g_report_fd >= 0) {                  // tell the process the user is waiting on, then let go of the streams it shares with us
        const ssize_t w = write(g_report_fd, &rc, sizeof(rc));
        (void)w;
        close(g_report_fd);
        const int devnull = open("/dev/null", O_RDWR);
        if (devnull >= 0) { dup2(devnull, 0); dup2(devnull, 1); dup2(devnull, 2); }
    }
    // `sketch` only: a profile / query that leaves through _exit hands the driver 29 GB of HBM and 0.5 GB of page-locked chunks to clean
    // up BEHIND the process — the next command's database load then ran at a third of its speed (profiles/r05_db_load_*.txt): those
    // two take the runtime's own way out.
    if (fast_exit() && argc >= 2 && !strcmp(argv[1], "sketch")) _exit(rc);
    return rc;
}
