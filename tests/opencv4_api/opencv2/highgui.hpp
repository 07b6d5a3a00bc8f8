/* OpenCV 4 highgui / imgcodecs declarations used by the facade's demo and by the reference's callers */
#ifndef SBM_TEST_OPENCV4_API_HIGHGUI_HPP
#define SBM_TEST_OPENCV4_API_HIGHGUI_HPP
#include "opencv2/core.hpp"
namespace cv {
enum ImreadModes { IMREAD_UNCHANGED = -1, IMREAD_GRAYSCALE = 0, IMREAD_COLOR = 1 };
Mat imread(const String& filename, int flags = IMREAD_COLOR);
bool imwrite(const String& filename, InputArray img, const std::vector<int>& params = std::vector<int>());
void imshow(const String& winname, InputArray mat);
int waitKey(int delay = 0);
void namedWindow(const String& winname, int flags = 1);
void destroyAllWindows();
} // namespace cv
#endif
