// compat/Frame_stereo.inl -- replacement BODIES for the two Frame members that sit on the hot path (reference src/Frame.cc).
//
// How to apply (maintainer, inside the ORB-SLAM2 tree; needs OpenCV, not compiled by this repository):
//   in src/Frame.cc, delete the bodies of Frame::ComputeStereoMatches (:880-1176) and, optionally, Frame::UndistortKeyPoints
//   (:770-825), and put   #include "Frame_stereo.inl"   in their place, inside namespace ORB_SLAM2.  Nothing else in
//   Frame.cc / Frame.h changes: ExtractORB (:468-481) already goes through compat/ORBextractor.h's operator().
//
// ComputeStereoMatches: the reference reads mpORBextractorLeft/Right->mvImagePyramid (:910, 1040, 1072, 1079), builds
// vRowIndices (:926-942), picks the best Hamming candidate per left keypoint (:990-1018), slides an 11x11 SAD window
// (:1040-1101), fits the parabola (:1121-1129) and cuts at 2.1 x the median SAD (:1160-1175).  orbx_stereo_match does all of
// that on the device against the pyramids the two extractor handles still hold from ExtractORB (fork semantics: padded
// levels, iniu = scaleduR0 - L - w, src/Frame.cc:1067), with the row index clamped (the reference's unchecked
// vRowIndices[yi] is out of bounds for coarse-level keypoints near the bottom edge).  mvuRight / mvDepth come back exactly as
// the reference leaves them (-1 where there is no match).

void Frame::ComputeStereoMatches()
{
    mvuRight = std::vector<float>(N, -1.0f);
    mvDepth = std::vector<float>(N, -1.0f);
    // nothing below reads mvImagePyramid: the extractors can stop copying the eight levels back after every frame
    // (compat/ORBextractor.h copies them by default so that the header alone stays a drop-in)
    mpORBextractorLeft->SetEagerPyramid(false);
    mpORBextractorRight->SetEagerPyramid(false);
    if (N == 0) return;
    static_assert(sizeof(cv::KeyPoint) == sizeof(orbx_keypoint), "cv::KeyPoint must be the 28-byte POD");
    int nmatches = 0;
    const orbx_status st = orbx_stereo_match(
        mpORBextractorLeft->handle(), mpORBextractorRight->handle(), /*frame_left*/ 0, /*frame_right*/ 0,
        reinterpret_cast<const orbx_keypoint *>(mvKeys.data()), mDescriptors.ptr<uint8_t>(), N,
        reinterpret_cast<const orbx_keypoint *>(mvKeysRight.data()), mDescriptorsRight.ptr<uint8_t>(), (int)mvKeysRight.size(),
        mb, mbf, mvuRight.data(), mvDepth.data(), &nmatches);
    if (st != ORBX_OK) throw std::runtime_error(orbx_last_error());
}

#ifdef ORBX_REPLACE_UNDISTORT
// Frame::UndistortKeyPoints (:770-825) = cv::undistortPoints(mat, mat, mK, mDistCoef, cv::Mat(), mK) on the keypoint
// coordinates; OpenCV 3.2 arithmetic (double precision, five fixed iterations) -- parity unpinned like every OpenCV-owned
// stage, so this replacement is opt-in.
void Frame::UndistortKeyPoints()
{
    if (mDistCoef.at<float>(0) == 0.0) { mvKeysUn = mvKeys; return; }
    const float cam[4] = {fx, fy, cx, cy};
    mvKeysUn.resize(N);
    if (N == 0) return;
    const int nd = (int)mDistCoef.total();
    if (orbx_undistort_keypoints(mpORBextractorLeft->handle(), reinterpret_cast<const orbx_keypoint *>(mvKeys.data()), N, cam,
                                 mDistCoef.ptr<float>(), nd, reinterpret_cast<orbx_keypoint *>(mvKeysUn.data())) != ORBX_OK)
        throw std::runtime_error(orbx_last_error());
}
#endif
