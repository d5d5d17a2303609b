// Command-line driver for the GPU-backed sampler: the --mcmc-slam / --multi-mcmc
// subset of the reference program (src/main.cpp), with the reference's flag
// names, seed derivation and replay order.  Everything else of that program
// (g2o, clustering, HMC, plotting, RMSE) is out of scope here.
//
//   slam_hip --dataset Plaza1 --dataset-dir tests/golden/plaza --mcmc-slam
//            --mcmc-steps 10 --mcmc-end-steps 1000 --sensor-range-stddev 1 --seed 42
//
// --dump-loader prints what the loader produced (no GPU needed).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <random>
#include <string>

#include "mcmc_slam_hip.hpp"
#include "simulator.hpp"

using control_model_type = planar_robot::velocity_slip_model;
// The observation model is a compile-time choice, as in the reference (src/main.hpp:21-22):
// range only by default, range and bearing with -DSLAM_HIP_RANGE_BEARING (binary slam_hip_rb).
#ifdef SLAM_HIP_RANGE_BEARING
using observation_model_type = planar_robot::range_bearing_model;
#else
using observation_model_type = planar_robot::range_only_model;
#endif
using slam_data_type = slam::slam_data<control_model_type, observation_model_type>;
using mcmc_slam_type = slam::mcmc_slam<control_model_type, observation_model_type>;
using multi_mcmc_type = slam::multi_mcmc<control_model_type, observation_model_type>;
using initialiser_type = slam::slam_initialiser<control_model_type, observation_model_type>;
using controller_type = planar_robot::waypoint_controller;
using sensor_type = planar_robot::landmark_sensor;
using simulator_type = simulator<control_model_type, observation_model_type, controller_type, sensor_type>;

namespace {

struct options {
  std::map<std::string, std::string> kv;
  bool has(const std::string& k) const { return kv.count(k) != 0; }
  std::string str(const std::string& k, const std::string& dflt) const { return has(k) ? kv.at(k) : dflt; }
  double num(const std::string& k, double dflt) const { return has(k) ? std::atof(kv.at(k).c_str()) : dflt; }
  unsigned int uint(const std::string& k, unsigned int dflt) const {
    return has(k) ? static_cast<unsigned int>(std::strtoul(kv.at(k).c_str(), nullptr, 10)) : dflt;
  }
};

const char* const kFlags[] = {"dataset", "dataset-dir", "mcmc-slam", "mcmc-steps", "mcmc-end-steps", "mcmc-slam-seed",
                              "multi-mcmc", "multi-mcmc-chains", "multi-mcmc-end-steps", "multi-mcmc-seed", "seed",
                              "init-seed", "sensor-range-stddev", "velmodel-alpha1", "velmodel-alpha2",
                              "velmodel-alpha3", "velmodel-alpha4", "velmodel-alpha5", "velmodel-alpha6",
                              "dump-loader", "dump-data", "device", "trace-every", "dataset-cache", "help",
                              // the simulated data source (used when no --dataset is given)
                              "waypoint-file", "landmark-file", "dt", "sensor-skip", "sim-seed", "waypoint-repetitions",
                              "waypoint-proximity", "sensor-range-max", "sensor-range-min", "robot-speed",
                              "robot-max-steering", "robot-max-steering-rate", "controller-paramP",
                              "controller-paramD", "controller-smoothing", "sensor-bearing-stddev", "split-slices",
                              "split-min-observations", "controls-out", "devices", "exchange"};
const char* const kSwitches[] = {"mcmc-slam", "multi-mcmc", "dump-loader", "help"};

options parse(int argc, char** argv) {
  options o;
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    if (a.rfind("--", 0) != 0) throw std::runtime_error("unexpected argument " + a);
    a = a.substr(2);
    bool known = false, is_switch = false;
    for (const char* f : kFlags) known = known || a == f;
    for (const char* f : kSwitches) is_switch = is_switch || a == f;
    if (!known) throw std::runtime_error("unknown option --" + a);
    if (is_switch)
      o.kv[a] = "1";
    else {
      if (i + 1 >= argc) throw std::runtime_error("option --" + a + " needs a value");
      o.kv[a] = argv[++i];
    }
  }
  return o;
}

}  // namespace

int main(int argc, char** argv) try {
  const options opt = parse(argc, argv);
  if (opt.has("help") || !(opt.has("dataset") || opt.has("waypoint-file") || opt.has("dataset-cache"))) {
    std::puts("usage: slam_hip (--dataset NAME [--dataset-dir DIR] | --waypoint-file F --landmark-file F [--dt X]\n"
              "                 [--sensor-skip N] [--sim-seed S] [--waypoint-repetitions X])\n"
              "                [--mcmc-slam --mcmc-steps N --mcmc-end-steps N]\n"
              "                [--multi-mcmc --multi-mcmc-chains C --multi-mcmc-end-steps N] [--seed S]\n"
              "                [--sensor-range-stddev X] [--velmodel-alpha1..6 X] [--dump-loader] [--dump-data PREFIX]\n"
              "                [--dataset-cache FILE] [--controls-out FILE] [--devices 0,1,..] [--exchange rccl|host|auto]");
    return opt.has("help") ? 0 : 2;
  }

  // Module seeds: six successive draws of random_source(seed), always made, in the order
  // sim, init, mcmc-slam, hmc, fastslam, multi-mcmc; an explicit --*-seed only replaces its own.
  const unsigned int seed = opt.has("seed") ? opt.uint("seed", 0) : std::random_device()();
  std::uint32_t draws[6];
  mcmcgpu_seed_sequence(seed, 6, draws);
  const unsigned int sim_seed = opt.uint("sim-seed", draws[0]);
  const unsigned int init_seed = opt.uint("init-seed", draws[1]);
  const unsigned int mcmc_slam_seed = opt.uint("mcmc-slam-seed", draws[2]);
  const unsigned int multi_mcmc_seed = opt.uint("multi-mcmc-seed", draws[5]);

  const double d = 3.14159265358979323846 / 180;
  const control_model_type::builder control_model_builder(
      opt.num("velmodel-alpha1", 0.01), opt.num("velmodel-alpha2", 0.0), opt.num("velmodel-alpha3", 1.0 * d * d),
      opt.num("velmodel-alpha4", 0.01), opt.num("velmodel-alpha5", 0.01 * d * d), opt.num("velmodel-alpha6", 0.001 * d));
#ifdef SLAM_HIP_RANGE_BEARING
  const observation_model_type::builder observation_model_builder(opt.num("sensor-range-stddev", 0.1),
                                                                  opt.num("sensor-bearing-stddev", 1.0 * d));
#else
  const observation_model_type::builder observation_model_builder(opt.num("sensor-range-stddev", 0.1));
#endif

  // --dataset-cache FILE: load the binary form if it exists, else parse the text logs and leave it behind
  std::shared_ptr<slam::dataset_impl<control_model_type, observation_model_type>> dataset;
  const std::string cache = opt.str("dataset-cache", "");
#ifdef SLAM_HIP_RANGE_BEARING
  if (opt.has("dataset") || !cache.empty())
    throw std::runtime_error("the Plaza logs and the dataset cache hold range-only readings; this binary simulates");
  {
#else
  if (!cache.empty() && std::ifstream(cache, std::ios::binary).good()) {
    dataset = slam::read_dataset_cache(cache);
  } else if (!opt.has("dataset")) {
#endif  // main.cpp:164-171 of the reference: no data set named -> simulate one
    controller_type controller;
    controller.waypoints = planar_robot::read_positions(opt.str("waypoint-file", ""));
    controller.speed = opt.num("robot-speed", controller.speed);
    controller.max_steering = opt.num("robot-max-steering", controller.max_steering);
    controller.max_steering_rate = opt.num("robot-max-steering-rate", controller.max_steering_rate);
    controller.proximity = opt.num("waypoint-proximity", controller.proximity);
    controller.repetitions = opt.num("waypoint-repetitions", controller.repetitions);
    controller.param_P = opt.num("controller-paramP", controller.param_P);
    controller.param_D = opt.num("controller-paramD", controller.param_D);
    controller.smoothing = opt.num("controller-smoothing", controller.smoothing);
    sensor_type sensor;
    if (opt.has("landmark-file")) sensor.landmarks = planar_robot::read_positions(opt.str("landmark-file", ""));
    sensor.max_range = opt.num("sensor-range-max", sensor.max_range);
    sensor.min_range = opt.num("sensor-range-min", sensor.min_range);
    dataset = std::make_shared<simulator_type>(sim_seed, control_model_builder, observation_model_builder, controller,
                                               sensor, opt.num("dt", 0.25), opt.uint("sensor-skip", 8));
#ifndef SLAM_HIP_RANGE_BEARING
    if (!cache.empty()) slam::write_dataset_cache(*dataset, cache);
  } else {
    // --controls-out FILE: the controls.txt the reference's loader writes into the data directory
    // (src/dataset.cpp:96-109), to a file of the caller's choice
    std::shared_ptr<slam::range_only_ground_truth> ground_truth;
    dataset = slam::read_range_only_data(opt.str("dataset-dir", "input"), opt.str("dataset", ""),
                                         opt.has("controls-out") ? &ground_truth : nullptr);
    if (ground_truth) slam::write_controls_txt(*dataset, *ground_truth, opt.str("controls-out", ""));
    if (!cache.empty()) slam::write_dataset_cache(*dataset, cache);
#endif
  }

  if (opt.has("dump-loader")) {
    const auto& obs = dataset->observations();
    std::printf("T %zu\nN %zu\n", std::size_t(dataset->current_timestep()), obs.size());
    for (std::size_t i = 0; i < obs.size(); ++i)
      if (i < 5 || i + 5 >= obs.size())
        std::printf("obs %zu %zu %zu %a %a\n", i, std::size_t(obs[i].first), std::size_t(obs[i].second.id),
                    obs[i].second.observation(0), obs[i].second.observation(observation_model_type::vector_dim - 1));
    double sm[3] = {0, 0, 0}, ss[3] = {0, 0, 0};
    for (std::size_t t = 0; t < dataset->current_timestep(); ++t) {
      const auto m = control_model_builder(dataset->control(slam::timestep_type(t)), dataset->timedelta(slam::timestep_type(t)));
      for (int i = 0; i < 3; ++i) {
        sm[i] += m.mean()(i);
        ss[i] += m.stddev()(i);
      }
    }
    std::printf("sum_ctrl_mean %a %a %a\nsum_ctrl_std %a %a %a\n", sm[0], sm[1], sm[2], ss[0], ss[1], ss[2]);
    return 0;
  }

  auto data = std::make_shared<slam_data_type>();
  auto init = std::make_shared<initialiser_type>(init_seed);
  data->add_listener(init);

  slam::gpu_capacity cap;
  cap.device = int(opt.uint("device", 0));
  cap.timesteps = std::uint32_t(std::size_t(dataset->current_timestep())) + 1;
  cap.observations = std::uint32_t(dataset->observations().size()) + 1;
  cap.features = 4096;
  cap.split_slices = opt.uint("split-slices", cap.split_slices);
  cap.split_min_observations = opt.uint("split-min-observations", cap.split_min_observations);
  // --devices 0,1,2,3: --multi-mcmc shards its chains over these GPUs (one handle each, RCCL between them);
  // --exchange rccl|host|auto: how the handles exchange (rccl also with one GPU: a communicator of one rank;
  // host: through host memory, the only way for handles that share a GPU)
  if (opt.has("devices")) {
    const std::string list = opt.str("devices", "");
    for (std::size_t at = 0; at <= list.size();) {
      const std::size_t comma = std::min(list.find(',', at), list.size());
      if (comma > at) cap.devices.push_back(std::atoi(list.substr(at, comma - at).c_str()));
      at = comma + 1;
    }
  }
  const std::string exchange = opt.str("exchange", "auto");
  if (exchange == "rccl")
    cap.exchange = slam::gpu_capacity::exchange_mode::rccl;
  else if (exchange == "host")
    cap.exchange = slam::gpu_capacity::exchange_mode::host;
  else if (exchange != "auto")
    throw std::runtime_error("--exchange takes rccl, host or auto");

  std::shared_ptr<mcmc_slam_type> mcmc_slam;
  std::shared_ptr<mcmc_slam_type::updater> mcmc_slam_updater;
  if (opt.has("mcmc-slam")) {
    mcmc_slam = std::make_shared<mcmc_slam_type>(data, mcmc_slam_seed, cap);
    mcmc_slam->set_initialiser(init);
    mcmc_slam_updater =
        std::make_shared<mcmc_slam_type::updater>(mcmc_slam, opt.uint("mcmc-steps", 0), opt.uint("mcmc-end-steps", 0));
    data->add_timestep_listener(mcmc_slam);
    data->add_timestep_listener(mcmc_slam_updater);
  }
  std::shared_ptr<multi_mcmc_type> multi_mcmc;
  if (opt.has("multi-mcmc")) {
    multi_mcmc = std::make_shared<multi_mcmc_type>(data, opt.uint("multi-mcmc-chains", 100),
                                                   opt.uint("multi-mcmc-end-steps", 0), multi_mcmc_seed, cap);
    data->add_timestep_listener(multi_mcmc);
  }

  const auto t0 = std::chrono::steady_clock::now();
  data->add_dataset(*dataset, control_model_builder, observation_model_builder);
  const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  std::printf("Wall Time: %.3f seconds\n\n", wall);

  if (mcmc_slam) {
    std::printf("MCMC-SLAM log likelihood: %.17g\n", mcmc_slam->get_log_likelihood());
    std::printf("MCMC-SLAM timesteps: %zu landmarks: %zu\n", std::size_t(mcmc_slam->current_timestep()),
                mcmc_slam->get_feature_map().size());
    if (opt.has("dump-data")) {  // traj_est / map_est of the reference's --dump-data, unaligned
      const std::string prefix = opt.str("dump-data", "");
      std::ofstream traj(prefix + "traj_est.txt"), map(prefix + "map_est.txt");
      for (slam::timestep_type t; t <= mcmc_slam->current_timestep(); ++t) {
        const auto s = mcmc_slam->get_state(t);
        traj << s.x() << ' ' << s.y() << '\n';
      }
      for (const auto& f : mcmc_slam->get_feature_map()) map << f.second.x() << ' ' << f.second.y() << '\n';
    }
  }
  if (multi_mcmc) {
    std::printf("Multi-MCMC-SLAM Chains: %u on %u handle(s), exchange through %s\n", multi_mcmc->num_chains(),
                multi_mcmc->num_handles(),
                multi_mcmc->exchanges_through_rccl() ? "RCCL" : multi_mcmc->num_handles() > 1 ? "host memory" : "nothing");
    {  // the best chain's estimate, as print_rmse / dump_data would read it (main.cpp:284-305)
      const auto& trajectory = multi_mcmc->get_trajectory();
      const auto last = multi_mcmc->get_state(multi_mcmc->current_timestep());
      std::printf("Multi-MCMC-SLAM best chain: %zu edges, final pose %.17g %.17g %.17g, %zu landmarks\n", trajectory.size(),
                  last.x(), last.y(), last.bearing(), multi_mcmc->get_feature_map().size());
    }
    std::printf("Multi-MCMC-SLAM log likelihood: %.17g\n", multi_mcmc->get_log_likelihood());
    // main.cpp:315-320 of the reference: the average over the chains and its likelihood
    const auto average = multi_mcmc->get_average();
    std::printf("Averaged-Multi-MCMC log likelihood: %.17g\n",
                slam::slam_log_likelihood<control_model_type, observation_model_type>(*data, *average, cap));
  }
  return 0;
} catch (const std::exception& e) {
  std::fprintf(stderr, "slam_hip: %s\n", e.what());
  return 1;
}
