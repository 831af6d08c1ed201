"""The parameter sets the reference ships (numbers transcribed from /root/reference/config/<name>/
{esio,esvio}.yaml and the camodocal calibration files next to them, PINHOLE model in all of them):
sensor sizes, feature budgets, publish rate, equalisation / motion-compensation switches and the
intrinsics + radial-tangential distortion of both event cameras and, for the ESVIO configs, both
frame cameras.  Data for tests/test_shipped_configs_gpu.py: every shipped parameterisation has to be
accepted by the handle and give the oracle's tracks."""


def _cam(fx, fy, cx, cy, k1, k2, p1, p2):
    return dict(fx=fx, fy=fy, cx=cx, cy=cy, k1=k1, k2=k2, p1=p1, p2=p2)


_DAVIS346_L = _cam(249.69341447817564, 248.41625664694038, 176.74240257052816, 129.47631010746218,
                   -0.3794794654640921, 0.15393049046270296, 0.0011400586965363895, -0.0019042695753031854)
_DAVIS346_R = _cam(258.61441518089174, 258.00363445501824, 178.44356547141308, 135.84792628403616,
                   -0.3864639588089853, 0.1707517912637013, -0.00046695742172563157, 0.0006610867041757214)
_DSEC_EV_L = _cam(553.4686750102932, 553.3994078799127, 346.65339162053317, 216.52092103243012,
                  -0.09356476362537607, 0.19445779814646236, 7.642434980998821e-05, 0.0019563864604273664)
_DSEC_EV_R = _cam(552.1819422959984, 551.4454720096484, 336.87432177064744, 226.32630571403274,
                  -0.026300, 0.037995, -0.000513, 0.000167)
_VECTOR_EV_L = _cam(327.32749, 327.46184, 304.97749, 235.37621, -0.031982, 0.041966, -0.000507, -0.001031)
_VECTOR_EV_R = _cam(327.48497, 327.55395, 318.53477, 230.96356, -0.026300, 0.037995, -0.000513, 0.000167)
_ECMD_EV_L = _cam(547.2703829559849, 545.4540498341149, 320.2935455165061, 241.843126203522,
                  -0.40575879021628114, 0.20616097747818823, -0.002622678645791178, -0.0008709741411368295)
_ECMD_EV_R = _cam(553.27973951009, 551.0377615505245, 334.78826704489654, 261.6924559105888,
                  -0.39688672555479626, 0.1861182570409217, -0.00021042147799614814, 0.0004195272663720832)
_MVSEC_L = _cam(226.38018519795807, 226.15002947047415, 173.6470807871759, 133.73271487507847,
                -0.048031442223833355, 0.011330957517194437, -0.055378166304281135, 0.021500973881459395)
_MVSEC_R = _cam(226.0181418548734, 225.7869434267677, 174.5433576736815, 124.21627572590607,
                -0.04846669832871334, 0.010092844338123635, -0.04293073765014637, 0.005194706897326005)

# name: event sensor, feature budget, freq, equalize, Do_motion_correction, event cameras, and for the
# ESVIO configs the frame camera (size, max_cnt_img, min_dist_img, cameras)
SHIPPED = {
    "esio": dict(ev=(346, 260), max_cnt=150, min_dist=10, freq=15, equalize=0, mc=0, ev_cams=(_DAVIS346_L, _DAVIS346_R)),
    "esio_DSEC": dict(ev=(640, 480), max_cnt=300, min_dist=10, freq=15, equalize=1, mc=0, ev_cams=(_DSEC_EV_L, _DSEC_EV_R)),
    "esvio": dict(ev=(346, 260), max_cnt=150, min_dist=10, freq=15, equalize=0, mc=0, ev_cams=(_DAVIS346_L, _DAVIS346_R),
                  img=(346, 260), max_cnt_img=150, min_dist_img=10, img_cams=(_DAVIS346_L, _DAVIS346_R)),
    "esvio_DSEC": dict(ev=(640, 480), max_cnt=100, min_dist=30, freq=10, equalize=0, mc=0, ev_cams=(_DSEC_EV_L, _DSEC_EV_R),
                       img=(1440, 1080), max_cnt_img=175, min_dist_img=40,
                       img_cams=(_cam(1150.8943600390282, 1150.8943600390282, 723.4334411621094, 572.102180480957, 0.0, 0.0, 0.0, 0.0),) * 2),
    "esvio_VECtor": dict(ev=(640, 480), max_cnt=150, min_dist=10, freq=10, equalize=0, mc=1, ev_cams=(_VECTOR_EV_L, _VECTOR_EV_R),
                         img=(1224, 1024), max_cnt_img=200, min_dist_img=20,
                         img_cams=(_cam(886.191073, 886.591633, 610.578911, 514.59271, -0.315760, 0.104955, 0.000320, -0.000156),
                                   _cam(887.804282, 888.04815, 616.177573, 514.712952, -0.311523, 0.09641, 0.000623, -0.000375))),
    "esvio_ecmd": dict(ev=(640, 480), max_cnt=200, min_dist=20, freq=10, equalize=0, mc=0, ev_cams=(_ECMD_EV_L, _ECMD_EV_R),
                       img=(1920, 1200), max_cnt_img=200, min_dist_img=30,
                       img_cams=(_cam(1088.6223477169553, 1083.9062438787385, 978.7220682606473, 584.9866756115756,
                                      -0.14356644984564232, 0.0802205318952682, -0.0008883818469204232, -0.000527072013337785),
                                 _cam(1060.3789912939371, 1055.9222603091423, 966.4501997389414, 590.8132971250201,
                                      -0.15843011501758728, 0.11138739987426229, 0.0009368301746198988, -0.00039117819198317166))),
    "esvio_mvsec_flying": dict(ev=(346, 260), max_cnt=150, min_dist=10, freq=15, equalize=0, mc=1, ev_cams=(_MVSEC_L, _MVSEC_R),
                               img=(346, 260), max_cnt_img=150, min_dist_img=10, img_cams=(_MVSEC_L, _MVSEC_R)),
}
# (esvio_VECtor_small_scale ships the same numbers as esvio_VECtor)
