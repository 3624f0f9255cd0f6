// mcmc_main.cpp -- the reference's command line (src/mcmc_main.cc) re-hosted on the HIP engine.
//
// Same flags (mcmc_main.cc:55-93), same validation messages and exit codes (:99-239), same initial
// partition rules (:241-326), same stdout contract: the label vector through output_vec (trailing blank,
// newline), "acceptance ratio" and summary() on clog (:483-485).  Boost.program_options is replaced by a
// small parser with the same surface (long/short names, `--opt=value`, multitoken options).
// Extra flags: --chains, --device, --devices, --rng {mt19937-compat,philox}, --gen_seed, --csr_cache, --reorder, --marginalize.
// The agglomerative drivers (:349-451) run through bisbm_agg_merge.  --merge starts at one block per node: while
// KA + KB > 256 the library runs its wide mode (two-byte labels, generic kernel), up to about 14 000 blocks (bisbm_check_shape).
// Negative diffs (agg_split) run through the same call (blockmodel.cc:110-117).
#include <chrono>
#include <limits>
#include <cmath>
#include <numeric>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <thread>
#include <atomic>
#include <vector>

#include "bisbm.hpp"

using namespace bisbm_host;

namespace {

struct option_spec {
    const char* long_name;
    char short_name;  // 0 = none
    int kind;         // 0 = flag, 1 = single value, 2 = multitoken
};

const option_spec kOptions[] = {
    {"edge_list_path", 'e', 1}, {"membership_path", 0, 1}, {"mb", 0, 2},          {"n", 'n', 2},
    {"types", 'y', 2},          {"burn_in", 'b', 1},       {"sampling_steps", 't', 1},
    {"sampling_frequency", 'f', 1}, {"bisbm_partition", 'z', 2}, {"uni", 0, 0},
    {"cooling_schedule", 'c', 1}, {"cooling_schedule_kwargs", 'a', 2}, {"steps_await", 'x', 1},
    {"epsilon", 'E', 1},        {"randomize", 'r', 0},     {"merge", 'g', 0},      {"nature", 'u', 0},
    {"seed", 'd', 1},           {"help", 'h', 0},
    // engine extras
    {"chains", 0, 1},           {"device", 0, 1},          {"devices", 0, 1},          {"rng", 0, 1},          {"gen_seed", 0, 1},
    {"csr_cache", 0, 0},        {"reorder", 0, 0},         {"marginalize", 0, 0},
};

const option_spec* find_long(const std::string& name) {
    for (auto const& o : kOptions)
        if (name == o.long_name) return &o;
    return nullptr;
}
const option_spec* find_short(char c) {
    for (auto const& o : kOptions)
        if (o.short_name && o.short_name == c) return &o;
    return nullptr;
}
bool looks_like_option(const char* s) {
    if (s[0] != '-' || s[1] == '\0') return false;
    if (s[1] == '-') return true;
    return !(s[1] >= '0' && s[1] <= '9') && s[1] != '.';  // "-3" / "-.5" are values
}

using var_map_t = std::map<std::string, std::vector<std::string>>;

bool parse_command_line(int argc, char const* argv[], var_map_t& vm, std::string& err) {
    for (int i = 1; i < argc; ++i) {
        std::string tok = argv[i];
        const option_spec* spec = nullptr;
        std::string inline_value;
        bool has_inline = false;
        if (tok.rfind("--", 0) == 0) {
            const size_t eq = tok.find('=');
            const std::string name = tok.substr(2, eq == std::string::npos ? std::string::npos : eq - 2);
            spec = find_long(name);
            if (!spec) {
                err = "unrecognised option '" + tok + "'";
                return false;
            }
            if (eq != std::string::npos) {
                inline_value = tok.substr(eq + 1);
                has_inline = true;
            }
        } else if (tok.size() >= 2 && tok[0] == '-') {
            spec = find_short(tok[1]);
            if (!spec) {
                err = "unrecognised option '" + tok + "'";
                return false;
            }
            if (tok.size() > 2) {
                inline_value = tok.substr(2);
                has_inline = true;
            }
        } else {
            err = "too many positional options have been specified on the command line";
            return false;
        }
        auto& values = vm[spec->long_name];
        if (spec->kind == 0) {
            values.push_back("");
            continue;
        }
        if (has_inline) values.push_back(inline_value);
        if (spec->kind == 1) {
            if (!has_inline) {
                if (i + 1 >= argc) {
                    err = std::string("the required argument for option '--") + spec->long_name + "' is missing";
                    return false;
                }
                values.push_back(argv[++i]);
            }
        } else {
            while (i + 1 < argc && !looks_like_option(argv[i + 1])) values.push_back(argv[++i]);
            if (values.empty()) {
                err = std::string("the required argument for option '--") + spec->long_name + "' is missing";
                return false;
            }
        }
    }
    return true;
}

uint_vec_t to_uints(const std::vector<std::string>& v) {
    uint_vec_t out;
    for (auto const& s : v) out.push_back((unsigned)std::strtoul(s.c_str(), nullptr, 10));
    return out;
}

void print_help(const char* argv0) {
    std::clog << "MCMC algorithms for the bipartiteSBM (final output only)\n";
    std::clog << "Usage:\n  " << argv0 << " [--option_1=value] [--option_s2=value] ...\n";
    std::clog << "Options:\n"
                 "  -e [ --edge_list_path ] arg           Path to edge list file.\n"
                 "  --membership_path arg                 Path to membership file.\n"
                 "  --mb arg                              Memberships given on the command line.\n"
                 "  -n [ --n ] arg                        Block sizes vector.\n"
                 "  -y [ --types ] arg                    Block types vector (NA NB).\n"
                 "  -b [ --burn_in ] arg (=1000)          Burn-in time (parsed, unused: mcmc_main.cc:61).\n"
                 "  -t [ --sampling_steps ] arg (=1000)   Length of the annealing process, in MH steps.\n"
                 "  -f [ --sampling_frequency ] arg (=10) (parsed, unused: mcmc_main.cc:64).\n"
                 "  -z [ --bisbm_partition ] arg          bipartite number of blocks to be inferred.\n"
                 "  --uni                                 (parsed, unused).\n"
                 "  -c [ --cooling_schedule ] arg (=abrupt_cool)\n"
                 "                                        exponential, linear, logarithmic, constant, abrupt_cool.\n"
                 "  -a [ --cooling_schedule_kwargs ] arg  Arguments of the cooling schedule (floats).\n"
                 "  -x [ --steps_await ] arg (=1000)      Stop after x steps without a new minimum.\n"
                 "  -E [ --epsilon ] arg (=1)             epsilon of the smart proposal.\n"
                 "  -r [ --randomize ]                    Randomize initial block state.\n"
                 "  -g [ --merge ]                        Start from one block per node and merge down to -z KA KB.\n"
                 "  -u [ --nature ]                       With --merge: merge until a type has fewer than sqrt(2E)/2 blocks.\n"
                 "  -d [ --seed ] arg                     Seed of the mt19937 engine (clock if absent).\n"
                 "  -h [ --help ]                         Produce this help message.\n"
                 "Engine:\n"
                 "  --chains arg (=1)                     Independent chains; the labels of the chain with the lowest\n"
                 "                                        description length are printed.\n"
                 "  --device arg (=0)                     HIP device ordinal.\n"
                 "  --devices arg                         Comma-separated HIP device ordinals: the --chains are spread over them\n"
                 "                                        (contiguous ranges, one handle; results do not depend on the split).\n"
                 "  --rng arg                             mt19937-compat (the reference's draw sequence: the default when -d\n"
                 "                                        is given) or philox (the production chain: the default otherwise).\n"
                 "  --gen_seed arg (=seed+1)              Seed of the reference's hidden second engine (blockmodel.hh:18).\n"
                 "  --reorder                             Renumber the nodes for memory locality before the run (ids without\n"
                 "                                        structure); labels are read and printed in the caller's numbering.\n"
                 "                                        The run is a different, equally valid chain than without the flag.\n"
                 "  --marginalize                         The marginalization mode of README.md:49-94 (the reference parses -b and\n"
                 "                                        -f and drops them): T = 1, -b burn-in steps, then -t steps with a\n"
                 "                                        sample every -f steps (whole sweeps, at least one between samples);\n"
                 "                                        prints every node's most frequent block over samples and chains.\n"
                 "  --csr_cache                           Keep a binary CSR beside the edge list (<path>.bisbm_csr, checked\n"
                 "                                        against the file's size and mtime); the text file stays the input.\n";
}

}  // namespace

int main(int argc, char const* argv[]) {
    var_map_t var_map;
    std::string err;
    if (!parse_command_line(argc, argv, var_map, err)) {
        std::cerr << err << "\n";
        return 1;
    }
    auto count = [&](const char* k) { return var_map.count(k) ? var_map[k].size() : (size_t)0; };
    auto single = [&](const char* k, const char* dflt) { return count(k) ? var_map[k].back() : std::string(dflt); };

    if (count("help") > 0 || argc == 1) {  // mcmc_main.cc:99-105
        print_help(argv[0]);
        return 0;
    }
    if (count("edge_list_path") == 0) {
        std::cerr << "edge_list_path is required (-e flag)\n";
        return 1;
    }
    size_t NA = 0, NB = 0;
    uint_vec_t types_init;
    uint_vec_t y = to_uints(var_map["types"]);
    if (count("types") == 0) {
        std::cerr << "types is required for bisbm mode (-y flag)\n";
        return 1;
    } else if (y.size() != 2) {
        std::cerr << "Number of types must be equal to 2!\n";
        return 1;
    } else {
        NA = y[0];
        NB = y[1];
        types_init.assign(NA + NB, 0);
        for (size_t i = NA; i < NA + NB; ++i) types_init[i] = 1;
    }
    const std::string cooling_schedule = single("cooling_schedule", "abrupt_cool");
    const size_t sampling_steps = std::strtoull(single("sampling_steps", "1000").c_str(), nullptr, 10);
    const size_t steps_await = std::strtoull(single("steps_await", "1000").c_str(), nullptr, 10);
    const double epsilon = std::strtod(single("epsilon", "1").c_str(), nullptr);
    float_vec_t kwargs(2, 0);
    if (count("cooling_schedule_kwargs") == 0) {  // defaults, mcmc_main.cc:134-153
        if (cooling_schedule == "exponential") {
            kwargs[0] = 1;
            kwargs[1] = 0.99f;
        }
        if (cooling_schedule == "linear") {
            kwargs[0] = (float)(sampling_steps + 1);
            kwargs[1] = 1;
        }
        if (cooling_schedule == "logarithmic") {
            kwargs[0] = 1;
            kwargs[1] = 1;
        }
        if (cooling_schedule == "constant") kwargs[0] = 1;
        if (cooling_schedule == "abrupt_cool") kwargs[0] = (float)steps_await;
    } else {  // checks, mcmc_main.cc:155-218
        auto const& a = var_map["cooling_schedule_kwargs"];
        kwargs.assign(std::max<size_t>(2, a.size()), 0);
        for (size_t i = 0; i < a.size(); ++i) kwargs[i] = std::strtof(a[i].c_str(), nullptr);
        if (cooling_schedule == "exponential") {
            if (kwargs[0] <= 0) {
                std::cerr << "Invalid cooling schedule argument for linear schedule: T_0 must be grater than 0.\n";
                std::cerr << "Passed value: T_0=" << kwargs[0] << "\n";
                return 1;
            }
            if (kwargs[1] <= 0 || kwargs[1] >= 1) {
                std::cerr << "Invalid cooling schedule argument for exponential schedule: alpha must be in ]0,1[.\n";
                std::cerr << "Passed value: alpha=" << kwargs[1] << "\n";
                return 1;
            }
        } else if (cooling_schedule == "linear") {
            if (kwargs[0] <= 0) {
                std::cerr << "Invalid cooling schedule argument for linear schedule: T_0 must be grater than 0.\n";
                std::cerr << "Passed value: T_0=" << kwargs[0] << "\n";
                return 1;
            }
            if (kwargs[1] <= 0 || kwargs[1] > kwargs[0]) {
                std::cerr << "Invalid cooling schedule argument for linear schedule: eta must be in ]0, T_0].\n";
                std::cerr << "Passed value: T_0=" << kwargs[0] << ", eta=" << kwargs[1] << "\n";
                return 1;
            }
            if (kwargs[1] * sampling_steps > kwargs[0]) {
                std::cerr << "Invalid cooling schedule argument for linear schedule: eta * sampling_steps must be "
                             "smaller or equal to T_0.\n";
                std::cerr << "Passed value: eta*sampling_steps=" << kwargs[1] * sampling_steps << ", T_0=" << kwargs[0]
                          << "\n";
                return 1;
            }
        } else if (cooling_schedule == "logarithmic") {
            if (kwargs[0] <= 0) {
                std::cerr << "Invalid cooling schedule argument for logarithmic schedule: c must be greater than 0.\n";
                std::cerr << "Passed value: c=" << kwargs[0] << "\n";
                return 1;
            }
            if (kwargs[1] <= 0) {
                std::cerr << "Invalid cooling schedule argument for logarithmic schedule: d must be greater than 0.\n";
                std::cerr << "Passed value: d=" << kwargs[1] << "\n";
                return 1;
            }
        } else if (cooling_schedule == "constant") {
            if (kwargs[0] <= 0) {
                std::cerr << "Invalid cooling schedule argument for constant schedule: temperature must be greater "
                             "than 0.\n";
                std::cerr << "Passed value: T=" << kwargs[0] << "\n";
                return 1;
            }
        } else if (cooling_schedule == "abrupt_cool") {
            if (kwargs[0] <= 0) {
                std::cerr << "Invalid cooling schedule argument for abrupt_cool schedule: tau must be larger than 0. \n";
                std::cerr << "Passed value: tau=" << kwargs[0] << "\n";
                return 1;
            }
        } else {
            std::cerr << "Invalid cooling schedule. Options are exponential, linear, logarithmic, abrupt_cool.\n";
            return 1;
        }
    }
    // mcmc_main.cc:219-226 tests var_map.count("epsilon"), which is always 1 because of the default value
    bool randomize = count("randomize") > 0;
    const bool merge = count("merge") > 0, nature = count("nature") > 0;
    size_t seed;
    if (count("seed") == 0)
        seed = (size_t)std::chrono::high_resolution_clock::now().time_since_epoch().count();  // :236-239
    else
        seed = std::strtoull(single("seed", "0").c_str(), nullptr, 10);

    // ---- initial partition, mcmc_main.cc:241-326 ----
    uint_vec_t memberships_init, n = to_uints(var_map["n"]), z = to_uints(var_map["bisbm_partition"]);
    uint_vec_t mb = to_uints(var_map["mb"]);
    size_t N = 0, KA = 0, KB = 0;
    bool prepared = false;
    if (count("membership_path") != 0) {
        std::clog << "Loading nodes' membership from membership_path.\n";
        if (!load_memberships(memberships_init, single("membership_path", ""))) {
            std::clog << "WARNING: error in loading memberships, read memberships from block sizes\n";
        } else {
            randomize = false;
            unsigned max_n_ka = 0, max_n_kb = 0;
            for (size_t i = 0; i < memberships_init.size(); ++i) {
                if (i < y[0] && memberships_init[i] > max_n_ka) max_n_ka = memberships_init[i];
                if (memberships_init[i] > max_n_kb) max_n_kb = memberships_init[i];
            }
            KA = max_n_ka + 1;
            KB = max_n_kb - max_n_ka;
            prepared = true;
            N = memberships_init.size();
            std::clog << " ---- read membership from file! ---- \n";
        }
    } else if (count("mb") > 0) {
        size_t accu = 0;
        for (auto it : n) accu += it;
        if (mb.size() != accu) {
            std::cerr << "[error] input vector size of memberships is different from the number of nodes \n";
            output_vec(mb, std::cerr);
            std::cerr << "#mb = " << mb.size() << "; while #nodes = " << accu << ". \n";
            return 1;
        }
        memberships_init = mb;
        if (z.size() < 2) {
            std::cerr << "number of partitions is required (-z flag)\n";
            return 1;
        }
        KA = z[0];
        KB = z[1];
        N = memberships_init.size();
        prepared = true;
    }
    if (!prepared) {
        if (count("n") == 0) {
            std::cerr << "n is required (-n flag) if one does not specify the membership of nodes\n";
            return 1;
        }
        for (size_t r = 0; r < n.size(); ++r)
            for (size_t i = 0; i < n[r]; ++i) memberships_init.push_back((unsigned)r);
        if (z.size() < 2) {
            std::cerr << "number of partitions is required (-z flag)\n";
            return 1;
        }
        KA = z[0];
        KB = z[1];
        N = memberships_init.size();
    }
    if (memberships_init.size() != types_init.size()) {  // :328-333
        std::cerr << memberships_init.size() << ", " << types_init.size() << '\n';
        std::cerr << "Types do not sum to the number of vertices!\n";
        return 1;
    }

    // ---- graph, mcmc_main.cc:335-339 ----
    adj_list_t adj_list_loaded;
    if (count("csr_cache")) {  // same arrays as the two calls below, kept in a binary file beside the text
        if (!load_adj_cached(adj_list_loaded, single("edge_list_path", ""), N, true)) {
            std::cerr << "[error] cannot read the edge list, or it names a node id >= " << N << "\n";
            return 1;
        }
    } else {
        edge_list_t edge_list;
        load_edge_list(edge_list, single("edge_list_path", ""));
        adj_list_loaded = edge_to_adj(edge_list, N);
    }
    // --reorder: the engine works on a renumbered graph; memberships go in and come out in the caller's numbering
    std::vector<uint32_t> new_id;
    if (count("reorder")) {
        new_id = locality_order(adj_list_loaded, NA);
        adj_list_loaded = permute_adj(adj_list_loaded, new_id);
        uint_vec_t moved(memberships_init.size());
        for (size_t v = 0; v < memberships_init.size(); ++v) moved[new_id[v]] = memberships_init[v];
        memberships_init.swap(moved);
    }
    auto emit_labels = [&](const uint_vec_t& engine_labels) {
        if (new_id.empty()) {
            output_vec<uint_vec_t>(engine_labels, std::cout);
            return;
        }
        uint_vec_t mine(engine_labels.size());
        for (size_t v = 0; v < mine.size(); ++v) mine[v] = engine_labels[new_id[v]];
        output_vec<uint_vec_t>(mine, std::cout);
    };
    const adj_list_t& adj_list = adj_list_loaded;

    // K implied by the initial labels vs. requested (mcmc_main.cc:406-419)
    size_t ka = 0, kb = 0;
    for (size_t t = 0; t < NA + NB; ++t) {
        if (types_init[t] == 0 && memberships_init[t] > ka)
            ka = memberships_init[t];
        else if (types_init[t] == 1 && memberships_init[t] > kb)
            kb = memberships_init[t];
    }
    kb -= ka;
    ka += 1;
    engine_options opt;
    opt.n_chains = (uint32_t)std::strtoul(single("chains", "1").c_str(), nullptr, 10);
    opt.device = std::atoi(single("device", "0").c_str());
    if (count("devices")) {
        const std::string list = single("devices", "");
        for (size_t pos = 0; pos <= list.size();) {
            const size_t comma = std::min(list.find(',', pos), list.size());
            const std::string item = list.substr(pos, comma - pos);
            char* end = nullptr;
            const long d = std::strtol(item.c_str(), &end, 10);
            if (item.empty() || *end != '\0' || d < 0) {
                std::cerr << "Invalid --devices. A comma-separated list of device ordinals, e.g. 0,1,2,3.\n";
                return 1;
            }
            opt.devices.push_back((int)d);
            pos = comma + 1;
        }
        if (opt.devices.size() > opt.n_chains) {
            std::cerr << "--devices lists " << opt.devices.size() << " devices for --chains " << opt.n_chains << ": every device needs a chain.\n";
            return 1;
        }
        opt.device = opt.devices[0];
    }
    // Without -d the reference seeds its engines from the clock and random_device (mcmc_main.cc:242, blockmodel.hh:17-18): there
    // is no draw sequence to reproduce, so the production chain (Philox) runs.  With -d the default is the verification mode
    // that reproduces the reference's own mt19937 sequence for that seed -- an order of magnitude slower per step.
    const bool seed_given = count("seed") != 0;
    const std::string rng = single("rng", seed_given ? "mt19937-compat" : "philox");
    if (!count("rng") && seed_given)
        std::clog << "rng: mt19937-compat (-d given: the reference's draw sequence for this seed; --rng philox runs the production chain, "
                     "about ten times faster per step)\n";
    if (rng != "mt19937-compat" && rng != "philox") {
        std::cerr << "Invalid --rng. Options are mt19937-compat, philox.\n";
        return 1;
    }
    opt.rng_mode = rng == "philox" ? BISBM_RNG_PHILOX : BISBM_RNG_MT19937_COMPAT;
    opt.seed = seed;
    opt.gen_seed = count("gen_seed") ? std::strtoull(single("gen_seed", "0").c_str(), nullptr, 10) : seed + 1;

    // ---- agglomerative drivers, mcmc_main.cc:349-451 ----
    const double sigma = 1.01;  // :349
    const float_vec_t agg_merge_kwargs(1, 0.f);
    auto print_best = [&](blockmodel_t& blockmodel, bool with_k) {
        uint32_t best = 0;
        if (opt.n_chains > 1) {
            const std::vector<double> dl = blockmodel.entropy_all();
            for (uint32_t c = 1; c < opt.n_chains; ++c)
                if (dl[c] < dl[best]) best = c;
            std::clog << "chains " << opt.n_chains << ", printing chain " << best << "\n";
        }
        blockmodel.summary(best);
        if (with_k) std::cout << blockmodel.get_KA() << " " << blockmodel.get_KB() << " ";  // :401-403
        emit_labels(*blockmodel.get_memberships(best));
    };
    // one stage per pair of the plan: merge, then a greedy sweep except after the last stage (:380-396, :425-444)
    auto staged_merges = [&](blockmodel_t& blockmodel, metropolis_hasting& algorithm, const std::vector<int>& ka_s,
                             const std::vector<int>& kb_s) -> bool {
        for (size_t i = 0; i + 1 < ka_s.size(); ++i) {
            blockmodel.agg_merge(-(ka_s[i + 1] - ka_s[i]), -(kb_s[i + 1] - kb_s[i]), 10);
            if (i != ka_s.size() - 2) {
                if (cooling_schedule != "abrupt_cool") {
                    std::cerr << "Only abrupt cooling annealing is supported.";
                    return false;
                }
                algorithm.anneal(blockmodel, &abrupt_cool_schedule, agg_merge_kwargs, (NA + NB) * 1, steps_await);
            }
        }
        return true;
    };
    if (merge) {
        // the start is one block per node (:350-353): ask the library whether it serves that many blocks (its wide mode ends at
        // about 14 000) before the expensive part starts
        if (NA + NB > 65535 || bisbm_check_shape((uint32_t)NA, (uint32_t)NB, opt.rng_mode) != BISBM_OK) {
            std::cerr << "[error] --merge starts from one block per node (" << NA + NB << " blocks): "
                      << (NA + NB > 65535 ? "block labels are at most two bytes on the device" : bisbm_last_error(nullptr))
                      << ". Start from an initial partition of fewer blocks (-n / --mb / --membership_path with a larger -z "
                         "than wanted is merged down the same way, mcmc_main.cc:419-450).\n";
            return 3;
        }
        try {
            std::iota(memberships_init.begin(), memberships_init.end(), 0);  // every node its own block (:350)
            if (nature && opt.n_chains > 1) {
                // agg_merge(engine, diff, nm) lets every run end with its own (Ka,Kb) (blockmodel.cc:208-271), and in this
                // driver every run also has its own stage sizes (ceil of ITS block count) and its own last stage: --nature
                // --chains N therefore runs every chain in a handle of its own with its global chain id (same streams as in
                // one handle), and prints the best.  (One handle does serve chains of
                // different shapes -- bisbm_agg_merge_total -- but applies one diff to all of them per call.)
                // The runs are independent handles: up to eight of them are in flight at a time, each on a host thread of its own
                // (its handle has its own stream; with --devices they take turns over the listed devices).
                if (cooling_schedule != "abrupt_cool" &&  // (the reference notices inside its first merge stage, :370-376)
                    std::min(NA, NB) >= (size_t)std::ceil(std::sqrt(2. * (double)(adj_list.col.size() / 2)) / 2)) {
                    std::cerr << "Only abrupt cooling annealing is supported.";
                    return 1;
                }
                struct run_result {
                    double dl = std::numeric_limits<double>::infinity();
                    uint_vec_t labels;
                    size_t ka = 0, kb = 0;
                    std::string error;
                };
                std::vector<run_result> runs(opt.n_chains);
                auto run_one = [&](uint32_t c) {
                    try {
                        engine_options one = opt;
                        one.n_chains = 1;
                        one.first_chain_id = opt.first_chain_id + c;
                        if (!opt.devices.empty()) one.device = opt.devices[c % opt.devices.size()];  // (--devices: the runs take turns)
                        one.devices.clear();
                        blockmodel_t blockmodel(memberships_init, types_init, NA + NB, NA, NB, epsilon, &adj_list, one);
                        blockmodel.init_bisbm();
                        metropolis_hasting algorithm;
                        size_t tKA = NA, tKB = NB, tGroups = NA + NB;
                        const size_t ceiling = (size_t)std::ceil(std::sqrt(2. * blockmodel.get_num_edges()) / 2);
                        while (tKA >= ceiling && tKB >= ceiling) {  // :357-377
                            blockmodel.agg_merge((int)std::ceil(tGroups * (sigma - 1) / sigma), 10);
                            tKA = blockmodel.get_KA();
                            tKB = blockmodel.get_KB();
                            tGroups = tKA + tKB;
                            algorithm.anneal(blockmodel, &abrupt_cool_schedule, agg_merge_kwargs, (NA + NB) * 1, steps_await);
                        }
                        algorithm.anneal(blockmodel, &abrupt_cool_schedule, kwargs, sampling_steps, steps_await);  // :398
                        runs[c].dl = blockmodel.entropy_all()[0];
                        runs[c].labels = *blockmodel.get_memberships(0);
                        runs[c].ka = blockmodel.get_KA();
                        runs[c].kb = blockmodel.get_KB();
                    } catch (const std::exception& e) {
                        runs[c].error = e.what();
                    }
                };
                {
                    const uint32_t width = std::min<uint32_t>(opt.n_chains, 8u);
                    std::vector<std::thread> pool;
                    std::atomic<uint32_t> next{0};
                    for (uint32_t t = 0; t < width; ++t)
                        pool.emplace_back([&] {
                            for (uint32_t c = next++; c < opt.n_chains; c = next++) run_one(c);
                        });
                    for (auto& t : pool) t.join();
                }
                double best_dl = std::numeric_limits<double>::infinity();
                uint_vec_t best_labels;
                size_t best_ka = 0, best_kb = 0, best_chain = 0;
                for (uint32_t c = 0; c < opt.n_chains; ++c) {
                    if (!runs[c].error.empty()) throw std::runtime_error(runs[c].error);
                    if (runs[c].dl < best_dl) {  // (the first of equals, as when the runs went one after the other)
                        best_dl = runs[c].dl;
                        best_labels = runs[c].labels;
                        best_ka = runs[c].ka;
                        best_kb = runs[c].kb;
                        best_chain = c;
                    }
                }
                std::clog << "chains " << opt.n_chains << ", printing chain " << best_chain << "\n";
                std::clog << "(Ka, Kb) = (" << best_ka << ", " << best_kb << ") \n";  // summary(), blockmodel.cc:748-751
                std::clog << "entropy: " << best_dl << "\n";
                std::cout << best_ka << " " << best_kb << " ";  // :401-403
                emit_labels(best_labels);
                return 0;
            }
            blockmodel_t blockmodel(memberships_init, types_init, NA + NB, NA, NB, epsilon, &adj_list, opt);
            blockmodel.init_bisbm();
            metropolis_hasting algorithm;
            if (nature) {  // :354-376
                size_t tKA = NA, tKB = NB, tGroups = NA + NB;
                const size_t ceiling = (size_t)std::ceil(std::sqrt(2. * blockmodel.get_num_edges()) / 2);
                while (tKA >= ceiling && tKB >= ceiling) {
                    blockmodel.agg_merge((int)std::ceil(tGroups * (sigma - 1) / sigma), 10);
                    tKA = blockmodel.get_KA();
                    tKB = blockmodel.get_KB();
                    tGroups = tKA + tKB;
                    if (cooling_schedule != "abrupt_cool") {
                        std::cerr << "Only abrupt cooling annealing is supported.";
                        return 1;
                    }
                    algorithm.anneal(blockmodel, &abrupt_cool_schedule, agg_merge_kwargs, (NA + NB) * 1, steps_await);
                }
            } else {  // :377-397
                const auto plan = geospace((long)NA, (long)KA, (long)NB, (long)KB, sigma);
                if (!staged_merges(blockmodel, algorithm, plan.first, plan.second)) return 1;
            }
            algorithm.anneal(blockmodel, &abrupt_cool_schedule, kwargs, sampling_steps, steps_await);  // :398
            print_best(blockmodel, nature);
        } catch (const std::exception& e) {
            std::cerr << e.what() << "\n";
            return 3;
        }
        return 0;
    }
    if (ka != KA || kb != KB) {  // the initial partition has other block counts than asked for (:419-450)
        try {
            int diff_a = (int)ka - (int)KA, diff_b = (int)kb - (int)KB;
            blockmodel_t blockmodel(memberships_init, types_init, ka + kb, ka, kb, epsilon, &adj_list, opt);
            blockmodel.init_bisbm();
            metropolis_hasting algorithm;
            if (diff_a >= 0 && diff_b >= 0) {
                const auto plan = geospace((long)(KA + diff_a), (long)KA, (long)(KB + diff_b), (long)KB, sigma);
                if (plan.first.size() == 1) blockmodel.agg_merge(diff_a, diff_b, 10);
                if (!staged_merges(blockmodel, algorithm, plan.first, plan.second)) return 1;
            } else {
                blockmodel.agg_merge(diff_a, diff_b, 100);  // :446 (negative diffs: agg_split)
            }
            algorithm.anneal(blockmodel, &abrupt_cool_schedule, kwargs, sampling_steps, steps_await);  // :447
            print_best(blockmodel, false);
        } catch (const std::exception& e) {
            std::cerr << e.what() << "\n";
            return 3;
        }
        return 0;
    }

    if (var_map.count("marginalize")) {
        // The marginalization mode README.md:49-94 describes and the reference's main never runs (-b and -f are parsed
        // and dropped, mcmc_main.cc:61-65): constant T = 1; -b burn-in steps; then -t sampling steps with a sample every
        // -f steps; every sample adds every chain's labels to a per-node histogram; the output line is each node's most
        // frequent block.  Steps are executed in whole sweeps, as anneal() does (duration / N), with at least one sweep
        // between samples.
        try {
            const size_t N = NA + NB;
            const size_t burn_in = std::strtoull(single("burn_in", "1000").c_str(), nullptr, 10);
            const size_t freq = std::strtoull(single("sampling_frequency", "10").c_str(), nullptr, 10);
            const size_t sweeps_between = std::max<size_t>(1, freq / N);
            const size_t n_samples = sampling_steps / (sweeps_between * N);
            blockmodel_t blockmodel(memberships_init, types_init, KA + KB, KA, KB, epsilon, &adj_list, opt);
            if (randomize)
                blockmodel.shuffle_bisbm();
            else
                blockmodel.init_bisbm();
            metropolis_hasting algorithm;
            const float_vec_t t1{1.f, 0.f};
            const size_t never = std::numeric_limits<size_t>::max();
            if (burn_in >= N) algorithm.anneal(blockmodel, &constant_schedule, t1, burn_in, never);
            blockmodel.marginals_reset();
            for (size_t sample = 0; sample < n_samples; ++sample) {
                algorithm.anneal(blockmodel, &constant_schedule, t1, sweeps_between * N, never);
                blockmodel.marginals_accumulate();
            }
            std::clog << "marginalize: burn-in " << burn_in / N << " sweeps, " << n_samples << " samples " << sweeps_between
                      << " sweep(s) apart, " << opt.n_chains << " chain(s) pooled\n";
            if (n_samples == 0) {
                std::cerr << "[error] --marginalize: -t " << sampling_steps << " steps hold no sample (" << sweeps_between * N
                          << " steps per sample)\n";
                return 1;
            }
            emit_labels(blockmodel.marginal_map_labels(NA));
        } catch (const std::exception& e) {
            std::cerr << e.what() << "\n";
            return 3;
        }
        return 0;
    }

    try {
        blockmodel_t blockmodel(memberships_init, types_init, KA + KB, KA, KB, epsilon, &adj_list, opt);  // :453
        if (randomize)
            blockmodel.shuffle_bisbm();
        else
            blockmodel.init_bisbm();
        metropolis_hasting algorithm;
        schedule_fn fn = cooling_schedule == "exponential"   ? &exponential_schedule
                         : cooling_schedule == "linear"      ? &linear_schedule
                         : cooling_schedule == "logarithmic" ? &logarithmic_schedule
                         : cooling_schedule == "constant"    ? &constant_schedule
                                                             : &abrupt_cool_schedule;
        algorithm.anneal(blockmodel, fn, kwargs, sampling_steps, steps_await);  // :462-482
        uint32_t best = 0;
        if (opt.n_chains > 1) {
            const std::vector<double> dl = blockmodel.entropy_all();
            for (uint32_t c = 1; c < opt.n_chains; ++c)
                if (dl[c] < dl[best]) best = c;
            std::clog << "chains " << opt.n_chains << ", printing chain " << best << "\n";
        }
        std::clog << "acceptance ratio " << algorithm.rates()[best] << "\n";  // :483
        blockmodel.summary(best);                                            // :484
        emit_labels(*blockmodel.get_memberships(best));  // :485
    } catch (const std::exception& e) {
        std::cerr << e.what() << "\n";
        return 3;
    }
    return 0;
}
