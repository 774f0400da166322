// inference.cpp — coverage-adjusted ANI statistics, kept on the host as the north_star prescribes (f64 throughout).
// Restates contain.rs:657-813 (statistics half of get_stats), :817-847 (ani_from_lambda), :849-898 (bootstrap_interval)
// and inference.rs:104-124,207-242 (mean, var, ratio_lambda).  Third-party arithmetic (statrs Poisson::cdf, fastrand) is
// restated from its published definition and is "parity unpinned" (DESIGN.md §2).
#include <algorithm>
#include <cmath>

#include "sylph_host.hpp"

namespace sylph_host {

// statrs 0.16.1 Poisson::cdf(x) = gamma_ur(x + 1, lambda): regularised upper incomplete gamma Q(a, x).
static double gamma_q(double a, double x) {
    if (x <= 0.0) return 1.0;
    const double gln = std::lgamma(a);
    if (x < a + 1.0) {
        double ap = a, sum = 1.0 / a, del = sum;
        for (int n = 0; n < 100000; n++) {
            ap += 1.0;
            del *= x / ap;
            sum += del;
            if (std::fabs(del) < std::fabs(sum) * 1e-17) break;
        }
        return 1.0 - sum * std::exp(-x + a * std::log(x) - gln);
    }
    const double tiny = 1e-300;
    double b = x + 1.0 - a, c = 1.0 / tiny, d = 1.0 / b, h = d;
    for (int i = 1; i < 100000; i++) {
        const double an = -(double)i * ((double)i - a);
        b += 2.0;
        d = an * d + b;
        if (std::fabs(d) < tiny) d = tiny;
        c = b + an / c;
        if (std::fabs(c) < tiny) c = tiny;
        d = 1.0 / d;
        const double del = d * c;
        h *= del;
        if (std::fabs(del - 1.0) < 1e-16) break;
    }
    return std::exp(-x + a * std::log(x) - gln) * h;
}
double poisson_cdf(double lambda, uint64_t x) { return gamma_q((double)x + 1.0, lambda); }

// inference.rs:207-242
std::optional<double> ratio_lambda(const std::vector<uint32_t>& full_covs, double min_count_correct) {
    size_t num_zero = 0;
    std::map<uint64_t, uint64_t> count_map;
    for (uint32_t x : full_covs) {
        if (x == 0) num_zero++;
        else count_map[x]++;
    }
    if (count_map.size() == 1) return std::nullopt;                          // :221
    if (full_covs.size() - num_zero < SAMPLE_SIZE_CUTOFF) return std::nullopt;   // :225
    // :228-230: sort (count, value) descending and take the first
    uint64_t best_count = 0, most_ind = 0;
    for (const auto& kv : count_map)
        if (kv.second > best_count || (kv.second == best_count && kv.first > most_ind)) { best_count = kv.second; most_ind = kv.first; }
    const auto it = count_map.find(most_ind + 1);
    if (it == count_map.end()) return std::nullopt;                          // :231
    const double count_p1 = (double)it->second, count = (double)best_count;
    if (count_p1 < min_count_correct || count < min_count_correct) return std::nullopt;   // :236
    return count_p1 / count * (double)(most_ind + 1);                        // :239
}

// contain.rs:817-847
std::optional<double> ani_from_lambda(std::optional<double> lambda, double k, const std::vector<uint32_t>& full_cov) {
    if (!lambda) return std::nullopt;
    size_t contain_count = 0;
    for (uint32_t x : full_cov) if (x != 0) contain_count++;
    const double adj_index = (double)contain_count / (1. - std::exp(-*lambda)) / (double)full_cov.size();
    const double ani = std::pow(adj_index, 1. / k);
    if (ani < 0. || std::isnan(ani)) return std::nullopt;
    return ani;
}

// fastrand 2.1.1 (third party): WyRand step + Lemire bounded integers; `fastrand::seed(7)` (contain.rs:854).
namespace {
struct WyRand {
    uint64_t s;
    uint64_t next() {
        s += 0x2d358dccaa6c78a5ULL;
        const unsigned __int128 t = (unsigned __int128)s * (unsigned __int128)(s ^ 0x8bb84b93962eacc9ULL);
        return (uint64_t)t ^ (uint64_t)(t >> 64);
    }
    uint64_t below(uint64_t n) {   // usize(..n)
        uint64_t r = next();
        unsigned __int128 m = (unsigned __int128)r * n;
        uint64_t hi = (uint64_t)(m >> 64), lo = (uint64_t)m;
        if (lo < n) {
            const uint64_t t = (0 - n) % n;
            while (lo < t) {
                r = next();
                m = (unsigned __int128)r * n;
                hi = (uint64_t)(m >> 64);
                lo = (uint64_t)m;
            }
        }
        return hi;
    }
};
}  // namespace

// contain.rs:849-898 (default estimator only)
static void bootstrap_interval(const std::vector<uint32_t>& covs_full, double k, const ContainArgs& args, AniResult& out) {
    WyRand rng{7};
    const size_t num_samp = covs_full.size();
    std::vector<double> res_ani, res_lambda;
    std::vector<uint32_t> rand_vec(num_samp);
    for (int it = 0; it < 100; it++) {
        for (size_t i = 0; i < num_samp; i++) rand_vec[i] = covs_full[rng.below(num_samp)];
        const auto lambda = ratio_lambda(rand_vec, args.min_count_correct);
        const auto ani = ani_from_lambda(lambda, k, rand_vec);
        if (ani && lambda && !std::isnan(*ani) && !std::isnan(*lambda)) { res_ani.push_back(*ani); res_lambda.push_back(*lambda); }
    }
    std::sort(res_ani.begin(), res_ani.end());
    std::sort(res_lambda.begin(), res_lambda.end());
    if (res_ani.size() < 50) return;
    const size_t suc = res_ani.size();
    out.ani_ci_lo = res_ani[suc * 5 / 100 - 1];
    out.ani_ci_hi = res_ani[suc * 95 / 100 - 1];
    out.lambda_ci_lo = res_lambda[suc * 5 / 100 - 1];
    out.lambda_ci_hi = res_lambda[suc * 95 / 100 - 1];
}

// contain.rs:657-813
std::optional<AniResult> stats_from_covs(const ContainArgs& args, std::vector<uint32_t> covs, size_t n_genome_kmers, uint64_t k,
                                         std::optional<size_t> kmers_lost) {
    if (covs.empty()) return std::nullopt;                                   // :654
    const size_t contain_count = covs.size();
    AniResult r;
    r.naive_ani = std::pow((double)contain_count / (double)n_genome_kmers, 1. / (double)k);   // :657-660
    std::sort(covs.begin(), covs.end());                                     // :661 (already sorted when they come from the GPU)
    const double median_cov = (double)covs[covs.size() / 2];                 // :663
    double max_cov = 1.7976931348623157e308;                                 // f64::MAX
    if (median_cov < 30.) {                                                  // :666-675
        for (size_t i = covs.size() / 2; i < covs.size(); i++) {
            if (poisson_cdf(median_cov, covs[i]) < CUTOFF_PVALUE) max_cov = (double)covs[i];
            else break;
        }
    }
    std::vector<uint32_t> full_covs(n_genome_kmers - contain_count, 0);      // :679
    for (uint32_t c : covs) if ((double)c <= max_cov) full_covs.push_back(c);   // :680-684
    uint32_t sum = 0;
    for (uint32_t x : full_covs) sum += x;                                   // iter().sum::<u32>()
    const double mean_cov = (double)sum / (double)full_covs.size();          // :689
    const double geq1_mean_cov = (double)sum / (double)covs.size();          // :690
    (void)mean_cov;
    std::optional<double> test_lambda;
    if (median_cov > MEDIAN_ANI_THRESHOLD) r.lambda_status = AdjustStatus::High;   // :692-694
    else {
        test_lambda = ratio_lambda(full_covs, args.min_count_correct);       // :695-713 (default estimator)
        r.lambda_status = test_lambda ? AdjustStatus::Lambda : AdjustStatus::Low;
        if (test_lambda) r.lambda = *test_lambda;
    }
    if (r.lambda_status == AdjustStatus::Lambda) r.final_est_cov = r.lambda;                    // :717-728
    else if (median_cov < MAX_MEDIAN_FOR_MEAN_FINAL_EST) r.final_est_cov = geq1_mean_cov;
    else r.final_est_cov = args.mean_coverage ? geq1_mean_cov : median_cov;
    std::optional<double> opt_lambda;                                        // :730-735
    if (r.lambda_status == AdjustStatus::Lambda) opt_lambda = r.final_est_cov;
    const auto opt_est_ani = ani_from_lambda(opt_lambda, (double)k, full_covs);   // :737
    r.final_est_ani = (!opt_lambda || !opt_est_ani || args.no_adj) ? r.naive_ani : *opt_est_ani;   // :739-744
    const double min_ani = args.minimum_ani ? *args.minimum_ani / 100. : (args.pseudotax ? MIN_ANI_P_DEF : MIN_ANI_DEF);
    if (r.final_est_ani < min_ani) return std::nullopt;                      // :746-764
    if (!args.no_ci && opt_lambda) bootstrap_interval(full_covs, (double)k, args, r);   // :766-773
    r.mean_cov = geq1_mean_cov;                                              // AniResult.mean_cov (:795)
    r.median_cov = median_cov;
    r.contain_count = contain_count;
    r.n_kmers = n_genome_kmers;
    r.kmers_lost = kmers_lost;
    return r;
}

}  // namespace sylph_host
