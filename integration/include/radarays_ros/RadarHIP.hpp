#ifndef RADARAYS_RADAR_HIP_HPP
#define RADARAYS_RADAR_HIP_HPP
// RadarHIP -- third backend of uos/radarays_ros next to RadarCPU (include/radarays_ros/RadarCPU.hpp:16-37) and RadarGPU
// (include/radarays_ros/RadarGPU.hpp:15-38): the per-azimuth loop of RadarCPU::simulate (src/radarays_ros/RadarCPU.cpp:155-548)
// runs on MI355X GPUs behind the C ABI of radarays_mi355.h (libradarays_mi355.so).
//
// This file and src/radarays_ros/RadarHIP.cpp are copied into the reference tree by integration/apply.py, which also applies
// the line-anchored insertions of integration/patches/*.json to src/radar_simulator.cpp and CMakeLists.txt.  They need ROS 1,
// cv_bridge and rmagine's math types (what Radar.hpp itself includes) and are therefore NOT compiled in the image this
// repository is developed in; its ROS-free twin (include/radarays_ros_amd/RadarHIP.hpp, same marshalling member for member)
// is compiled and tested on the GPU against the oracle.

#include "Radar.hpp"

#include <radarays_mi355.h>

#include <string>
#include <vector>

namespace radarays_ros
{

class RadarHIP : public Radar
{
public:
    using Base = Radar;

    // Same first five arguments as RadarCPU / RadarGPU (RadarCPU.hpp:21-28).  The map: rr_load_mesh_file reads what
    // rm::import_embree_map reads for the node (src/radar_simulator.cpp:149) -- .ply, .obj, .dae; object ids (the index into
    // `object_materials`) follow the scene's depth-first order -- an ASSUMPTION about rmagine's numbering that cannot be checked
    // here; private parameter ~hip_object_order (list of object names) renumbers them, and the second constructor takes any
    // numbering the caller already holds.  `devices`: the GPUs of this node the backend fans out over
    // (azimuth blocks, one RCCL gather per frame).  `build_on_gpu`: LBVH built on device 0 (a 10M-triangle map loads in 0.35 s
    // instead of 1.9 s; frames take 1.2x as long).
    RadarHIP(
        std::shared_ptr<ros::NodeHandle> nh_p,
        std::shared_ptr<tf2_ros::Buffer> tf_buffer,
        std::shared_ptr<tf2_ros::TransformListener> tf_listener,
        std::string map_frame,
        std::string sensor_frame,
        const std::string& map_file,
        const std::vector<int>& devices = std::vector<int>(1, 0),
        bool build_on_gpu = false
    );

    // For callers that already hold the triangles (e.g. assimp's arrays where rmagine is installed, which keeps rmagine's
    // own object numbering): verts [nv][3], faces [nf][3], face_object [nf] or empty.
    RadarHIP(
        std::shared_ptr<ros::NodeHandle> nh_p,
        std::shared_ptr<tf2_ros::Buffer> tf_buffer,
        std::shared_ptr<tf2_ros::TransformListener> tf_listener,
        std::string map_frame,
        std::string sensor_frame,
        const std::vector<float>& verts,
        const std::vector<uint32_t>& faces,
        const std::vector<uint32_t>& face_object,
        const std::vector<int>& devices = std::vector<int>(1, 0),
        bool build_on_gpu = false
    );

    virtual ~RadarHIP();
    RadarHIP(const RadarHIP&) = delete;
    RadarHIP& operator=(const RadarHIP&) = delete;

    // Radar.hpp:64
    virtual sensor_msgs::ImagePtr simulate(ros::Time stamp);

    // Offline generation: one image per pose, up to 64 (RR_MAX_BATCH) poses per set of launches (rr_multi_simulate_batch),
    // every other piece of state as simulate() would use it.
    std::vector<sensor_msgs::ImagePtr> simulateBatch(const std::vector<rm::Transform>& poses, ros::Time stamp);
    // The same with include_motion's per-azimuth poses: sweeps[f] holds the n_angles poses of frame f (one table row per
    // frame, rr_multi_set_motion_poses).
    std::vector<sensor_msgs::ImagePtr> simulateSweeps(const std::vector<std::vector<rm::Transform> >& sweeps, ros::Time stamp);

    // The gen_radar_image action (action/GenRadarImage.action, scripts/radaray_opti.py:170-211) for many RadarParams at
    // once: images and / or the PSNR of each against `real` (the optimiser's objective is its negative).
    bool simulateParamSets(const std::vector<RadarParams>& sets, ros::Time stamp, std::vector<sensor_msgs::ImagePtr>* images,
                           const sensor_msgs::Image* real = nullptr, std::vector<double>* psnr = nullptr);

protected:
    void init(const std::vector<float>& verts, const std::vector<uint32_t>& faces, const std::vector<uint32_t>& face_object,
              const std::vector<int>& devices, bool build_on_gpu);
    // marshals the protected state simulate() reads (Radar.hpp:66-105) into the rr_multi_set_* calls; false + warning on error
    bool pushState();
    // include_motion: the per-azimuth lookups of RadarCPU.cpp:190-196 for one sweep; skipped[a] != 0 where the reference
    // would `continue` (no transform yet).  Returns the number of azimuths that have a pose
    int lookupSweep(ros::Time stamp, std::vector<float>& sweep, std::vector<char>& skipped);
    sensor_msgs::ImagePtr wrap(const unsigned char* pixels, ros::Time stamp) const;
    sensor_msgs::ImagePtr fail();

    rr_multi* m_multi = nullptr;   // one rr_ctx per device + the RCCL communicator
    rr_ctx* m_ctx = nullptr;       // device 0: parameter batches, statistics
    int m_n_angles = 400;
    uint32_t m_beam_seed = 0;      // the seed the current m_waves_start was drawn with (parameter batches re-use it)
    // > 0: in motion mode azimuth a of a sweep is looked up at stamp - (1 - a / n_angles) * sweep_duration (Radar::updateTsm(stamp),
    // Radar.cpp:134-186) instead of "latest" -- the reference ties the motion to the wall-clock time its own loop takes
    // (one lookup + ros::spinOnce per column, RadarCPU.cpp:190-196,544-547), which a GPU frame of 0.6 ms no longer spans.
    // Private parameter ~hip_sweep_duration [s], default 0 = the reference's behaviour
    double m_sweep_duration = 0.0;
};

using RadarHIPPtr = std::shared_ptr<RadarHIP>;

} // namespace radarays_ros

#endif // RADARAYS_RADAR_HIP_HPP
