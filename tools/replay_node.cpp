// replay_node.cpp — C++ host harness that drives libesvio_fe.so the way the reference's ROS node
// drives FeatureTracker: `sync_process` + `handle_stereo_event` of
// feature_tracker/src/stereo_event_tracker_node.cpp:145-344,372-418 without ROS.  It replays a log
// of dvs_msgs/EventArray messages (left and right topics) and dumps, per published frame, the rows
// of the sensor_msgs/PointCloud the node would publish (node:273-329).
//
//   replay_node <log.esvb> <dump.bin> [key=value ...]
//
// keys: max_cnt min_dist freq equalize flow_back f_threshold f_ransac decay_ms filter_thr
//       mc (1: Do_motion_correction — the Motion_correction_value of node:192-254 is assembled from the
//       logged IMU / odometry messages and the batch goes through esvio_fe_track_event_mc; ahead = 0)
//       fx fy cx cy (the YAML's intrinsics the warp uses; default: the left camera's)
//       ahead (0: one batch in flight like the reference; 1..3: esvio_fe_set_next_batch replay mode)
//       lazy threads  (throughput options, results identical)   rccl (1: also all-gather every
//       published frame's records over a one-rank RCCL communicator, esvio_fe_exchange_tracks)
//
// Log format (little endian): "ESVB" u32 version (1 or 2) u32 width u32 height u32 n_messages, then
// per message: u8 kind u8 pad[3] u32 n f64 header_stamp and
//   kind 0 / 1  dvs_msgs/EventArray of the left / right camera: n x 16 B dvs_msgs::Event (esvio_fe_event)
//   kind 2      sensor_msgs/Imu (version 2): n = 6 f64 — angular_velocity xyz, linear_acceleration xyz
//   kind 3      nav_msgs/Odometry of the back end (version 2): n = 3 f64 — twist.twist.linear xyz
//   kind 4 / 5  sensor_msgs/Image (mono8) of the left / right camera (version 2): n = width*height bytes.
//               A log with images is replayed the way the reference's second node does it
//               (stereo_image_tracker_node.cpp: sync_process :210-250 pairs within 1 s,
//               handle_stereo_image :54-183 -> esvio_fe_track_image); the two nodes are separate
//               processes in the reference, so a log holds either events or images.
// Messages are delivered to the callbacks in log order; a left/right pair is handled as soon as its
// second message has arrived (the reference's spinner and sync_process threads interleave freely).
// Dump format: "ESVD" u32 n_frames, per frame: f64 stamp u8 restart_flag u8 published u8 pad[2]
// u32 n_rows, n_rows x 8 f32 (x_un, y_un, 1, id*2+cam, u, v, vx, vy).
//
// Build: g++ -O2 -std=c++17 tools/replay_node.cpp -Iinclude -Lesvio_amd -lesvio_fe
//        -Wl,-rpath,$PWD/esvio_amd -Wl,-rpath-link,/opt/rocm/lib -ldl -o tools/replay_node
#include <dlfcn.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <set>
#include <string>
#include <vector>

#include "../include/esvio_fe.h"

namespace {

struct EventArray {  // dvs_msgs::EventArray: header.stamp + events
  double stamp = 0;
  std::vector<esvio_fe_event> events;
};

struct ImuMsg {  // sensor_msgs::Imu: header.stamp, angular_velocity, linear_acceleration
  double stamp, w[3], a[3];
};
struct OdomMsg {  // nav_msgs::Odometry: header.stamp, twist.twist.linear
  double stamp, v[3];
};

struct Frame {
  double stamp;
  uint8_t restart, published;
  std::vector<float> rows;  // n x 8
};

double to_sec(const esvio_fe_event& e) { return (double)e.sec + 1e-9 * (double)e.nsec; }  // ros::Time::toSec

struct Node {
  // file-scope state of the reference node (node:29-47)
  esvio_fe_handle h = nullptr;
  esvio_fe_config cfg{};
  int FREQ = 15;
  bool first_image_flag = true, init_pub = false, PUB_THIS_FRAME = false;
  double first_image_time = 0, last_image_time = 0;
  int pub_count = 1;
  // motion compensation (node:33-34,48-50,102-125): the IMU and back-end odometry queues, the last
  // two velocities the back end reported and their times (file-scope objects: zero at start)
  bool Do_motion_correction = false;
  double K[4] = {0, 0, 0, 0};  // fx, fy, cx, cy of the YAML
  std::deque<ImuMsg> imu_buf;
  std::deque<OdomMsg> odom_buffer_;
  double last_imu_t = 0, t_pre = 0, t_cur = 0;
  float v_cur[3] = {0, 0, 0}, v_pre[3] = {0, 0, 0};
  // result buffers (what the node reads from FeatureTracker's public members, node:289-322)
  std::vector<int32_t> ids, track_cnt, ids_right;
  std::vector<float> cur_pts, cur_un_pts, pts_velocity, cur_right_pts, cur_un_right_pts, right_pts_velocity;
  esvio_fe_tracks tr{};
  std::vector<Frame> out;
  // throughput options
  int ahead = 0;
  bool lazy = false;
  void* comm = nullptr;  // ncclComm_t (rccl=1)
  std::vector<float> gathered;
  int exchanges = 0;

  void alloc() {
    const size_t M = (size_t)cfg.max_cnt;
    ids.resize(M); track_cnt.resize(M); ids_right.resize(M);
    for (auto* v : {&cur_pts, &cur_un_pts, &pts_velocity, &cur_right_pts, &cur_un_right_pts, &right_pts_velocity})
      v->resize(2 * M);
    tr.ids = ids.data(); tr.track_cnt = track_cnt.data(); tr.cur_pts = cur_pts.data();
    tr.cur_un_pts = cur_un_pts.data(); tr.pts_velocity = pts_velocity.data(); tr.ids_right = ids_right.data();
    tr.cur_right_pts = cur_right_pts.data(); tr.cur_un_right_pts = cur_un_right_pts.data();
    tr.right_pts_velocity = right_pts_velocity.data();
  }

  void imu_callback(const ImuMsg& m) {  // node:109-125
    if (m.stamp <= last_imu_t) return;  // "imu message in disorder!"
    last_imu_t = m.stamp;
    imu_buf.push_back(m);
  }
  void state_callback(const OdomMsg& m) { odom_buffer_.push_back(m); }  // node:102-107

  // The Motion_correction_value of node:195-252 as an esvio_fe_motion.  Where the reference leaves a
  // local uninitialised (State_ and temp_a without a new odometry message, omega_avg_ without an IMU
  // message at or after the batch's first event) the field is zero here: zero acceleration is below
  // a_motion_compensation_threshold, i.e. the batch is not warped.
  esvio_fe_motion motion_value(const EventArray& L) {
    esvio_fe_motion mo{};
    const double t_left_0 = to_sec(L.events[0]);
    mo.t1 = L.stamp;  // t_left_1 = event_left.header.stamp (trackEvent re-reads both times itself, :621-622)
    if (!imu_buf.empty()) {
      if (!odom_buffer_.empty()) {
        const OdomMsg o = odom_buffer_.front();
        odom_buffer_.pop_front();
        for (int i = 0; i < 3; i++) {
          mo.v[i] = o.v[i];      // State_[0..2] = temp_v (doubles)
          v_pre[i] = v_cur[i];   // Vector3f
          v_cur[i] = (float)o.v[i];
        }
        t_pre = t_cur;
        t_cur = o.stamp;
        for (int i = 0; i < 3; i++) mo.accel[i] = (float)((double)(v_cur[i] - v_pre[i]) / (t_cur - t_pre));
      }
      while (!imu_buf.empty() && imu_buf.front().stamp < t_left_0) imu_buf.pop_front();
      if (!imu_buf.empty())
        for (int i = 0; i < 3; i++) mo.omega[i] = (float)imu_buf.front().w[i];
    }
    for (int i = 0; i < 3; i++) mo.v_pre[i] = v_pre[i];
    mo.fx = K[0];  // detector.init(COL, ROW, fx, fy, cx, cy) (feature_tracker.cpp:616): the YAML's
    mo.fy = K[1];  // fx, fy, cx, cy (parameters.cpp:221-224)
    mo.cx = K[2];
    mo.cy = K[3];
    return mo;
  }

  // the frequency-control decision of node:177-188 for a frame at `t` — `commit` false: only look
  bool freq_rule(double t, bool commit) {
    const double rate = 1.0 * pub_count / (t - first_image_time);
    if (std::round(rate) <= FREQ) {
      if (commit && std::fabs(rate - FREQ) < 0.01 * FREQ) {
        first_image_time = t;
        pub_count = 0;
      }
      return true;
    }
    return false;
  }

  // the PointCloud of node:273-329 (stereo_image_tracker_node.cpp:113-168 is the same loop)
  void pack_rows(Frame& f) {
    std::set<int> hash_ids;
    for (int j = 0; j < tr.n_left; j++)
      if (track_cnt[j] > 1) {  // node:289
        hash_ids.insert(ids[j]);
        const float row[8] = {cur_un_pts[2 * j], cur_un_pts[2 * j + 1], 1.f, (float)(ids[j] * 2 + 0),
                              cur_pts[2 * j], cur_pts[2 * j + 1], pts_velocity[2 * j], pts_velocity[2 * j + 1]};
        f.rows.insert(f.rows.end(), row, row + 8);
      }
    for (int j = 0; j < tr.n_right; j++)
      if (hash_ids.count(ids_right[j])) {  // node:309
        const float row[8] = {cur_un_right_pts[2 * j], cur_un_right_pts[2 * j + 1], 1.f,
                              (float)(ids_right[j] * 2 + 1), cur_right_pts[2 * j], cur_right_pts[2 * j + 1],
                              right_pts_velocity[2 * j], right_pts_velocity[2 * j + 1]};
        f.rows.insert(f.rows.end(), row, row + 8);
      }
  }

  // handle_stereo_image (stereo_image_tracker_node.cpp:54-183): the same first-frame / discontinuity
  // / publish-rate rules, trackImage, the same PointCloud.  (EQUALIZE's CLAHE of both images, :92-96,
  // is the handle's `equalize` option.)
  int handle_image(const std::vector<uint8_t>& L, const std::vector<uint8_t>& R, double msg_timestamp) {
    Frame f{msg_timestamp, 0, 0, {}};
    if (first_image_flag) {  // :58-64
      first_image_flag = false;
      first_image_time = msg_timestamp;
      last_image_time = msg_timestamp;
      return 0;
    }
    if (msg_timestamp - last_image_time > 1.0 || msg_timestamp < last_image_time) {  // :66-78
      first_image_flag = true;
      last_image_time = 0;
      pub_count = 1;
      f.restart = 1;
      out.push_back(f);
      return 0;
    }
    last_image_time = msg_timestamp;
    PUB_THIS_FRAME = freq_rule(msg_timestamp, true);  // :81-91
    const int rc = esvio_fe_track_image(h, msg_timestamp, L.data(), R.empty() ? nullptr : R.data(),
                                        PUB_THIS_FRAME ? 1 : 0, &tr);  // :100
    if (rc) return rc;
    if (PUB_THIS_FRAME) {
      pub_count++;  // :113-115
      pack_rows(f);
      if (!init_pub) {  // :171-176
        init_pub = true;
        f.rows.clear();
      } else {
        f.published = 1;
      }
    }
    out.push_back(f);
    return 0;
  }

  // handle_stereo_event (node:145-344).  `next`: the batches that will follow (replay mode only).
  int handle(const EventArray& L, const EventArray& R, double msg_timestamp,
             const std::vector<std::pair<const EventArray*, const EventArray*>>& next) {
    Frame f{msg_timestamp, 0, 0, {}};
    if (L.events.empty()) return 0;  // node:150
    if (first_image_flag) {           // node:155-161
      first_image_flag = false;
      first_image_time = msg_timestamp;
      last_image_time = msg_timestamp;
      return 0;
    }
    if (msg_timestamp - last_image_time > 1.0 || msg_timestamp < last_image_time) {  // node:163-173
      first_image_flag = true;
      last_image_time = 0;
      pub_count = 1;
      f.restart = 1;  // pub_restart.publish(true): the tracker itself is left alone
      out.push_back(f);
      return 0;
    }
    last_image_time = msg_timestamp;
    PUB_THIS_FRAME = freq_rule(msg_timestamp, true);                 // node:177-188
    const double msg_timestamp_left = to_sec(L.events.back());       // node:190
    if (ahead > 0) {  // replay mode: announce the following batches with the PUB they will carry
      // (the rule reads timestamps only: run it forward on a copy of the counters)
      Node sim;
      sim.FREQ = FREQ;
      sim.first_image_time = first_image_time;
      sim.pub_count = pub_count + (PUB_THIS_FRAME ? 1 : 0);
      int k = 0;
      for (auto& nb : next) {
        const bool pub = sim.freq_rule(nb.first->stamp, true);
        if (pub) sim.pub_count++;
        if (k++ < announced_ahead) continue;  // announced by an earlier call
        const int rc = esvio_fe_set_next_batch(h, to_sec(nb.first->events.back()), nb.first->events.data(),
                                               nb.first->events.size(), nb.second->events.data(),
                                               nb.second->events.size(), ESVIO_FE_HOST, pub ? 1 : 0);
        if (rc) return rc;
        announced_ahead++;
      }
    }
    int rc;
    if (!Do_motion_correction) {  // node:192-193
      rc = esvio_fe_track_event(h, msg_timestamp_left, L.events.data(), L.events.size(), R.events.data(),
                                R.events.size(), ESVIO_FE_HOST, PUB_THIS_FRAME ? 1 : 0, &tr);
    } else {  // node:194-254
      const esvio_fe_motion mo = motion_value(L);
      rc = esvio_fe_track_event_mc(h, msg_timestamp_left, L.events.data(), L.events.size(), R.events.data(),
                                   R.events.size(), ESVIO_FE_HOST, PUB_THIS_FRAME ? 1 : 0, &mo, &tr);
    }
    if (rc) return rc;
    if (announced_ahead > 0) announced_ahead--;
    if (PUB_THIS_FRAME) {
      pub_count++;  // node:268
      pack_rows(f);
      if (comm) {  // the multi-GPU hand-off: every rank's records, here a communicator of one rank
        gathered.assign((size_t)2 * cfg.max_cnt * 8, 0.f);
        if ((rc = esvio_fe_exchange_tracks(h, comm, 1, gathered.data()))) return rc;
        // the gathered block must be this frame's PointCloud rows followed by padding rows (id -1)
        const size_t n = f.rows.size();
        if (std::memcmp(gathered.data(), f.rows.data(), n * 4) != 0 ||
            (n / 8 < (size_t)2 * cfg.max_cnt && gathered[n + 3] != -1.f)) {
          fprintf(stderr, "exchange_tracks: gathered block differs from the packed rows\n");
          return -100;
        }
        exchanges++;
      }
      if (!init_pub) {  // node:334-339: the first publishable frame is swallowed
        init_pub = true;
        f.rows.clear();
      } else {
        f.published = 1;
      }
    }
    out.push_back(f);
    return 0;
  }
  int announced_ahead = 0;
};

struct Message {  // one logged message of any kind
  int kind = 0;
  EventArray ev;  // kinds 0, 1
  ImuMsg imu{};   // kind 2
  OdomMsg odom{}; // kind 3
  double stamp = 0;            // kinds 4, 5
  std::vector<uint8_t> image;  // kinds 4, 5
};

bool read_log(const char* path, int* W, int* H, std::vector<Message>* msgs) {
  FILE* fp = fopen(path, "rb");
  if (!fp) return false;
  char magic[4];
  uint32_t hdr[4];
  if (fread(magic, 1, 4, fp) != 4 || std::memcmp(magic, "ESVB", 4) != 0 || fread(hdr, 4, 4, fp) != 4 ||
      (hdr[0] != 1 && hdr[0] != 2)) {
    fclose(fp);
    return false;
  }
  *W = (int)hdr[1];
  *H = (int)hdr[2];
  for (uint32_t i = 0; i < hdr[3]; i++) {
    uint8_t cam[4];
    uint32_t n;
    double stamp;
    if (fread(cam, 1, 4, fp) != 4 || fread(&n, 4, 1, fp) != 1 || fread(&stamp, 8, 1, fp) != 1) break;
    Message m;
    m.kind = (int)cam[0];
    if (m.kind <= 1) {
      m.ev.stamp = stamp;
      m.ev.events.resize(n);
      if (n && fread(m.ev.events.data(), 16, n, fp) != n) break;
    } else if (m.kind == 2 && n == 6 && hdr[0] >= 2) {
      double v[6];
      if (fread(v, 8, 6, fp) != 6) break;
      m.imu = ImuMsg{stamp, {v[0], v[1], v[2]}, {v[3], v[4], v[5]}};
    } else if (m.kind == 3 && n == 3 && hdr[0] >= 2) {
      double v[3];
      if (fread(v, 8, 3, fp) != 3) break;
      m.odom = OdomMsg{stamp, {v[0], v[1], v[2]}};
    } else if ((m.kind == 4 || m.kind == 5) && n == hdr[1] * hdr[2] && hdr[0] >= 2) {
      m.stamp = stamp;
      m.image.resize(n);
      if (fread(m.image.data(), 1, n, fp) != n) break;
    } else {
      fclose(fp);
      return false;
    }
    msgs->push_back(std::move(m));
  }
  fclose(fp);
  return true;
}

int write_dump(const char* path, Node& node, size_t n_pairs, size_t thrown) {
  FILE* fo = fopen(path, "wb");
  if (!fo) return 2;
  const uint32_t nf = (uint32_t)node.out.size();
  fwrite("ESVD", 1, 4, fo);
  fwrite(&nf, 4, 1, fo);
  size_t n_pub = 0, n_rows = 0;
  for (const Frame& f : node.out) {
    const uint8_t flags[4] = {f.restart, f.published, 0, 0};
    const uint32_t n = (uint32_t)(f.rows.size() / 8);
    fwrite(&f.stamp, 8, 1, fo);
    fwrite(flags, 1, 4, fo);
    fwrite(&n, 4, 1, fo);
    if (n) fwrite(f.rows.data(), 4, f.rows.size(), fo);
    n_pub += f.published;
    n_rows += n;
  }
  fclose(fo);
  printf("replay_node: %zu message pairs (%zu thrown), %u frames tracked or restarted, %zu published, %zu rows, "
         "%d RCCL exchanges\n", n_pairs, thrown, nf, n_pub, n_rows, node.exchanges);
  esvio_fe_destroy(node.h);
  return 0;
}

}  // namespace

int main(int argc, char** argv) {
  if (argc < 3) {
    fprintf(stderr, "usage: %s <log.esvb> <dump.bin> [key=value ...]\n", argv[0]);
    return 2;
  }
  std::map<std::string, double> kv = {{"max_cnt", 300}, {"min_dist", 10},   {"freq", 15},     {"equalize", 0},
                                      {"flow_back", 1}, {"f_threshold", 1}, {"f_ransac", 1},  {"decay_ms", 20},
                                      {"filter_thr", 0.01}, {"ahead", 0},   {"lazy", 0},      {"threads", 1},
                                      {"rccl", 0},          {"mc", 0},      {"lk_accum", 2}};
  for (int i = 3; i < argc; i++) {
    const char* eq = std::strchr(argv[i], '=');
    if (!eq) continue;
    kv[std::string(argv[i], eq - argv[i])] = atof(eq + 1);
  }
  int W = 0, H = 0;
  std::vector<Message> msgs;
  if (!read_log(argv[1], &W, &H, &msgs)) {
    fprintf(stderr, "cannot read %s\n", argv[1]);
    return 2;
  }
  Node node;
  esvio_fe_config& c = node.cfg;
  c.width = W; c.height = H;
  c.decay_ms = kv["decay_ms"]; c.ignore_polarity = 0; c.median_blur_kernel_size = 0;
  c.feature_filter_threshold = kv["filter_thr"]; c.ts_lk_threshold = 128.0;
  c.max_cnt = (int)kv["max_cnt"]; c.min_dist = (int)kv["min_dist"]; c.flow_back = (int)kv["flow_back"];
  c.equalize = (int)kv["equalize"]; c.f_threshold = kv["f_threshold"]; c.f_ransac = (int)kv["f_ransac"];
  c.lk_accum = (int)kv["lk_accum"];  // 2: calcOpticalFlowPyrLK's float sums in the x86 OpenCV build's order; 1: exact sums
  c.focal_length = 460; c.device = -1;
  for (int k = 0; k < 2; k++)  // the tests' synthetic calibration (esvio_amd/frontend.py make_config)
    c.cam[k] = esvio_fe_camera{0.9 * W, 0.9 * W, W / 2.0, H / 2.0, -0.05, 0.01, 1e-4, -2e-4};
  node.FREQ = (int)kv["freq"] == 0 ? 100 : (int)kv["freq"];  // parameters.cpp:278-279
  node.ahead = (int)kv["ahead"];
  node.lazy = kv["lazy"] != 0;
  node.Do_motion_correction = kv["mc"] != 0;
  if (node.Do_motion_correction && node.ahead > 0) {
    // (the library can take an announced batch's Motion_correction_value — esvio_fe_set_next_batch_mc —
    // but this harness assembles it from the IMU / odometry messages that arrive up to the batch, like
    // the node does, so it only exists when the batch is handled)
    fprintf(stderr, "mc=1 needs ahead=0: the Motion_correction_value of a batch is assembled when it is handled\n");
    return 2;
  }
  node.K[0] = kv.count("fx") ? kv["fx"] : c.cam[0].fx;
  node.K[1] = kv.count("fy") ? kv["fy"] : c.cam[0].fy;
  node.K[2] = kv.count("cx") ? kv["cx"] : c.cam[0].cx;
  node.K[3] = kv.count("cy") ? kv["cy"] : c.cam[0].cy;
  int rc = esvio_fe_create(&c, &node.h);
  if (rc) {
    fprintf(stderr, "esvio_fe_create failed: %d\n", rc);
    return 1;
  }
  node.alloc();
  if (node.lazy) esvio_fe_set_lazy_new_stereo(node.h, 1);
  if (kv["threads"] > 1) esvio_fe_set_host_threads(node.h, (int)kv["threads"]);
  if (kv["rccl"] != 0) {
    // a communicator of one rank (this box has one GPU); on a multi-GPU node every rank passes its own
    void* lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    struct UniqueId { char internal[128]; };
    auto get_id = lib ? (int (*)(UniqueId*))dlsym(lib, "ncclGetUniqueId") : nullptr;
    auto init_rank = lib ? (int (*)(void**, int, UniqueId, int))dlsym(lib, "ncclCommInitRank") : nullptr;
    UniqueId id;
    if (!get_id || !init_rank || get_id(&id) != 0 || init_rank(&node.comm, 1, id, 0) != 0) {
      fprintf(stderr, "RCCL communicator could not be created\n");
      return 1;
    }
  }

  bool image_log = false;
  for (const Message& m : msgs) image_log = image_log || m.kind == 4 || m.kind == 5;
  if (image_log) {
    // sync_process of the image node (stereo_image_tracker_node.cpp:210-250): pair when the stamps
    // are within a second, else throw the older one
    std::deque<const Message*> il, ir;
    size_t pairs_n = 0, thrown_n = 0;
    for (const Message& m : msgs) {
      if (m.kind != 4 && m.kind != 5) continue;
      (m.kind == 4 ? il : ir).push_back(&m);
      while (!il.empty() && !ir.empty()) {
        const double tl = il.front()->stamp, trr = ir.front()->stamp;
        if (tl <= trr - 1) {
          il.pop_front();
          thrown_n++;
        } else if (tl > trr + 1) {
          ir.pop_front();
          thrown_n++;
        } else {
          rc = node.handle_image(il.front()->image, ir.front()->image, tl);
          if (rc) {
            fprintf(stderr, "image pair %zu failed: %d %s\n", pairs_n, rc, esvio_fe_last_error(node.h));
            return 1;
          }
          pairs_n++;
          il.pop_front();
          ir.pop_front();
        }
      }
    }
    return write_dump(argv[2], node, pairs_n, thrown_n);
  }

  // sync_process (node:372-418) over the logged messages: two queues, pair when |dt| <= 0.2 s, else
  // throw the older one.  (The reference's queues hold one message and drop under load; a replay
  // has no load, so every logged message is considered.)
  std::deque<const EventArray*> ql, qr;
  std::vector<std::pair<const EventArray*, const EventArray*>> pairs;
  std::vector<size_t> ready_at;  // index of the message that completed the pair
  size_t thrown = 0;
  for (size_t mi = 0; mi < msgs.size(); mi++) {
    const Message& m = msgs[mi];
    if (m.kind > 1) continue;
    (m.kind == 0 ? ql : qr).push_back(&m.ev);
    while (!ql.empty() && !qr.empty()) {
      const double tl = ql.front()->stamp, trr = qr.front()->stamp;
      if (tl < trr - 0.2) {
        ql.pop_front();
        thrown++;
      } else if (tl > trr + 0.2) {
        qr.pop_front();
        thrown++;
      } else {
        pairs.emplace_back(ql.front(), qr.front());
        ready_at.push_back(mi);
        ql.pop_front();
        qr.pop_front();
      }
    }
  }
  size_t delivered = 0;  // IMU / odometry messages reach their callbacks in log order, a pair is
                         // handled when its second message has arrived
  for (size_t i = 0; i < pairs.size(); i++) {
    for (; delivered < ready_at[i]; delivered++) {
      if (msgs[delivered].kind == 2) node.imu_callback(msgs[delivered].imu);
      if (msgs[delivered].kind == 3) node.state_callback(msgs[delivered].odom);
    }
    if (pairs[i].first->events.empty()) continue;  // node:408
    std::vector<std::pair<const EventArray*, const EventArray*>> next;
    for (size_t k = i + 1; k < pairs.size() && (int)next.size() < node.ahead; k++) {
      if (pairs[k].first->events.empty()) break;
      next.push_back(pairs[k]);
    }
    // (a discontinuity ahead would change the plan: replay mode only looks ahead inside a
    // continuous stretch)
    for (size_t k = 0; k < next.size(); k++) {
      const double tp = k ? next[k - 1].first->stamp : pairs[i].first->stamp, t = next[k].first->stamp;
      if (t - tp > 1.0 || t < tp) {
        next.resize(k);
        break;
      }
    }
    if (node.first_image_flag) next.clear();  // the first frame of a stretch is not tracked at all
    rc = node.handle(*pairs[i].first, *pairs[i].second, pairs[i].first->stamp, next);
    if (rc) {
      fprintf(stderr, "frame %zu failed: %d %s\n", i, rc, rc > -100 ? esvio_fe_last_error(node.h) : "");
      return 1;
    }
  }
  return write_dump(argv[2], node, pairs.size(), thrown);
}
