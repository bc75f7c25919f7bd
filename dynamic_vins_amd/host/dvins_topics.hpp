// dvins_topics.hpp — the ROS1 topic surface of the reference node, as data: what a ROS translation unit wrapped around dvins_node's loop has to subscribe to and
// advertise so that the node drops in.  No ROS headers are needed to read it (there is no ROS in this image); a wrapper instantiates
// nh.subscribe / nh.advertise<...> from these rows.  tests/test_node.py::test_topic_table_matches_the_reference re-reads the reference sources it cites.
#pragma once
namespace dynamic_vins {
struct TopicIn  { const char* topic_or_config_key; bool from_config; const char* msg_type; int queue; const char* callback; const char* condition; };
struct TopicOut { const char* topic; const char* msg_type; const char* publisher; };
// SystemCallBack::SystemCallBack (utils/io/system_call_back.cpp:18-35).  from_config: the topic name is the value of that key in the config file.
static const TopicIn kSubscriptions[] = {
    { "image0_topic", true, "sensor_msgs/Image", 100, "Img0Callback", "" },
    { "image1_topic", true, "sensor_msgs/Image", 100, "Img1Callback", "" },
    { "image0_segmentation_topic", true, "sensor_msgs/Image", 100, "Seg0Callback", "cfg::is_input_seg (VIODE, naive / dynamic)" },
    { "image1_segmentation_topic", true, "sensor_msgs/Image", 100, "Seg1Callback", "cfg::is_input_seg (VIODE, naive / dynamic)" },
    { "imu_topic", true, "sensor_msgs/Imu", 2000, "ImuCallback", "tcpNoDelay" },
    { "/vins_restart", false, "std_msgs/Bool", 100, "RestartCallback", "" },
    { "/vins_terminal", false, "std_msgs/Bool", 100, "TerminalCallback", "" },
    { "/vins_imu_switch", false, "std_msgs/Bool", 100, "ImuSwitchCallback", "" },
    { "/vins_cam_switch", false, "std_msgs/Bool", 100, "CamSwitchCallback", "" },
};
// PublisherMap::Pub<T> / PubPointCloud / PubMarkers / PubImage call sites (utils/io/visualization.cpp:85-633, system/main.cpp:316, estimator/estimator.cpp:1726);
// all node-private ("~"), queue 1000 (publisher_map.h:50-52)
static const TopicOut kPublications[] = {
    { "imu_propagate", "nav_msgs/Odometry", "PubLatestOdometry" },
    { "odometry", "nav_msgs/Odometry", "PubOdometry" },
    { "path", "nav_msgs/Path", "PubOdometry" },
    { "key_poses", "visualization_msgs/Marker", "PubKeyPoses" },
    { "camera_pose", "nav_msgs/Odometry", "PubCameraPose" },
    { "camera_pose_visual", "visualization_msgs/MarkerArray", "PubCameraPose" },
    { "point_cloud", "sensor_msgs/PointCloud", "PubPointCloud" },
    { "margin_cloud", "sensor_msgs/PointCloud", "PubPointCloud" },
    { "extrinsic", "nav_msgs/Odometry", "PubTF" },
    { "keyframe_pose", "nav_msgs/Odometry", "PubKeyframe" },
    { "keyframe_point", "sensor_msgs/PointCloud", "PubKeyframe" },
    { "instance_marker", "visualization_msgs/MarkerArray", "PubInstances / PubPredictBox3D / PubGroundTruthBox3D" },
    { "lines", "visualization_msgs/MarkerArray", "PubLines" },
    { "instance_point_cloud", "sensor_msgs/PointCloud2", "PubInstancePointCloud" },
    { "stereo_point_cloud", "sensor_msgs/PointCloud2", "PubStereoPointCloud" },
    { "scene_vec", "visualization_msgs/MarkerArray", "PubSceneVec" },
    { "image_track", "sensor_msgs/Image", "FeatureTrack (system/main.cpp:316)" },
    { "top_view", "sensor_msgs/Image", "Estimator::Output (estimator.cpp:1726)" },
};
}
