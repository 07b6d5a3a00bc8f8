// demo.cpp — command-line driver over the drop-in line2Dup::Detector facade, in the
// shape of the reference's test.cpp:angle_test (test.cpp:262-420) without highgui:
//   demo match <templ_fmt> <class_id> <image.ppm|.pgm> <threshold> <num_features> [pad]
//       readClasses + (optional zero padding, crop to multiples of 16) + match; prints the matches
//   demo train <image.ppm> <mask.pgm> <num_features> <n_rot> <angle_step> <templ_fmt> <class_id> [info.yaml]
//       addTemplate (angle 0) + addTemplate_rotate for the remaining angles + writeClasses
//   demo convert <in_fmt> <class_id> <out_fmt>
//       readClasses + writeClasses (no GPU needed)
//   demo scale_train <image.ppm> <num_features> <scale_lo> <scale_hi> <scale_step> <templ_fmt> <class_id> <info.yaml>
//       test.cpp:scale_test("train") (test.cpp:170-203): shapeInfo_producer::src_of / mask_of (cv::resize) + addTemplate
//   demo nms <templ_fmt> <class_id> <image> <threshold> <num_features> [pad]
//       test.cpp:noise_test (test.cpp:455-491): match, boxes from templ[0].width/height, NMSBoxes(boxes, scores, 0, 0.5f)
//   demo latency <templ_fmt> <class_id> <image> <threshold> <num_features> <n> [pad]
//       n timed calls of detector.match(img, threshold, ids) on one cv::Mat, as a test.cpp-style caller sees them
//       also: the same call with the frame buffer pinned (Detector::pinBuffer) and matchBatch from pinned host frames
//   demo batch <templ_fmt> <class_id> <image> <threshold> <num_features> <n_frames> <pad> <devices>
//       MI355X extensions: matchBatch / matchAsync+wait over n_frames shifted copies of the image against per-frame match(),
//       and match() with setDevices(<devices>, e.g. 0,0) against the single-context match()
//   demo instance <config.yaml> <image> <threshold>
//       Detector::getInstance(path) (line2Dup.cpp:1366-1393) + match over the classes the config lists
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <thread>
#include <vector>

#include "../../include/line2Dup.h"
#include "../../include/nms.hpp"

using namespace cv;

static int usage()
{
    fprintf(stderr, "usage: demo match|train ... (see demo.cpp)\n");
    return 2;
}

int main(int argc, char** argv)
{
    try {
        if (argc < 2) return usage();
        const std::string mode = argv[1];
        if (mode == "match") {
            if (argc < 7) return usage();
            const std::string fmt = argv[2], class_id = argv[3], path = argv[4];
            const float threshold = (float)atof(argv[5]);
            const int num_features = atoi(argv[6]);
            const int pad = argc > 7 ? atoi(argv[7]) : 0;
            line2Dup::Detector detector(num_features, {4, 8});
            std::vector<std::string> ids{class_id};
            detector.readClasses(ids, fmt);
            Mat test_img = imread(path, IMREAD_UNCHANGED);
            if (test_img.empty()) { fprintf(stderr, "cannot read %s\n", path.c_str()); return 1; }
            // test.cpp:344-353: pad, then crop to multiples of 16
            Mat padded(test_img.rows + 2 * pad, test_img.cols + 2 * pad, test_img.type(), Scalar::all(0));
            Mat inner = padded(Rect(pad, pad, test_img.cols, test_img.rows));
            test_img.copyTo(inner);
            const int stride = 16;
            Mat img = padded(Rect(0, 0, stride * (padded.cols / stride), stride * (padded.rows / stride))).clone();
            std::vector<line2Dup::Match> matches = detector.match(img, threshold, ids);
            printf("matches %zu templates %d image %dx%dx%d\n", matches.size(), detector.numTemplates(), img.rows, img.cols, img.channels());
            for (const auto& m : matches) {
                uint32_t bits;
                memcpy(&bits, &m.similarity, 4);
                printf("%d %d %u %s %d\n", m.x, m.y, bits, m.class_id.c_str(), m.template_id);
            }
            return 0;
        }
        if (mode == "train") {
            if (argc < 9) return usage();
            Mat img = imread(argv[2], IMREAD_UNCHANGED), mask = imread(argv[3], IMREAD_GRAYSCALE);
            const int num_features = atoi(argv[4]), n_rot = atoi(argv[5]);
            const float angle_step = (float)atof(argv[6]);
            const std::string fmt = argv[7], class_id = argv[8];
            if (img.empty() || mask.empty()) { fprintf(stderr, "cannot read inputs\n"); return 1; }
            line2Dup::Detector detector(num_features, {4, 8});
            shape_based_matching::shapeInfo_producer shapes(img, mask);
            std::vector<shape_based_matching::shapeInfo_producer::Info> infos;
            int first_id = detector.addTemplate(shapes.src, class_id, shapes.mask); // test.cpp:304
            if (first_id < 0) { fprintf(stderr, "addTemplate failed\n"); return 1; }
            infos.emplace_back(0.f, 1.f);
            for (int k = 1; k <= n_rot; ++k) { // test.cpp:310-312
                float angle = 0.f;
                for (int j = 0; j < k; ++j) angle += angle_step; // same float accumulation as produce_infos
                detector.addTemplate_rotate(class_id, first_id, angle, {shapes.src.cols / 2.0f, shapes.src.rows / 2.0f});
                infos.emplace_back(angle, 1.f);
            }
            detector.writeClasses(fmt);
            if (argc > 9) shape_based_matching::shapeInfo_producer::save_infos(infos, argv[9]);
            printf("trained %d templates\n", detector.numTemplates());
            return 0;
        }
        if (mode == "scale_train") {
            if (argc < 10) return usage();
            Mat img = imread(argv[2], IMREAD_UNCHANGED);
            if (img.empty()) { fprintf(stderr, "cannot read %s\n", argv[2]); return 1; }
            const int num_feature = atoi(argv[3]);
            const std::string fmt = argv[7], class_id = argv[8];
            line2Dup::Detector detector(num_feature, {4, 8});
            shape_based_matching::shapeInfo_producer shapes(img);
            shapes.scale_range = {(float)atof(argv[4]), (float)atof(argv[5])};
            shapes.scale_step = (float)atof(argv[6]);
            shapes.produce_infos();
            std::vector<shape_based_matching::shapeInfo_producer::Info> infos_have_templ;
            for (auto& info : shapes.infos) {
                // the call of test.cpp:185-186, argument for argument (in this fork the 4th parameter is `sscale`)
                int templ_id = detector.addTemplate(shapes.src_of(info), class_id, shapes.mask_of(info), int(num_feature * info.scale));
                if (templ_id != -1) infos_have_templ.push_back(info);
            }
            detector.writeClasses(fmt);
            shape_based_matching::shapeInfo_producer::save_infos(infos_have_templ, argv[9]);
            printf("trained %d templates of %zu infos\n", detector.numTemplates(), shapes.infos.size());
            return 0;
        }
        if (mode == "nms") {
            if (argc < 7) return usage();
            const std::string fmt = argv[2], class_id = argv[3], path = argv[4];
            const float threshold = (float)atof(argv[5]);
            const int num_features = atoi(argv[6]);
            const int pad = argc > 7 ? atoi(argv[7]) : 0;
            line2Dup::Detector detector(num_features, {4, 8});
            std::vector<std::string> ids{class_id};
            detector.readClasses(ids, fmt);
            Mat test_img = imread(path, IMREAD_UNCHANGED);
            if (test_img.empty()) { fprintf(stderr, "cannot read %s\n", path.c_str()); return 1; }
            Mat padded(test_img.rows + 2 * pad, test_img.cols + 2 * pad, test_img.type(), Scalar::all(0));
            test_img.copyTo(padded(Rect(pad, pad, test_img.cols, test_img.rows)));
            const int stride = 16;
            Mat img = padded(Rect(0, 0, stride * (padded.cols / stride), stride * (padded.rows / stride))).clone();
            auto matches = detector.match(img, threshold, ids);
            std::vector<Rect> boxes;
            std::vector<float> scores;
            std::vector<int> idxs;
            for (auto match : matches) { // test.cpp:478-490
                Rect box;
                box.x = match.x;
                box.y = match.y;
                auto templ = detector.getTemplates(class_id, match.template_id);
                box.width = templ[0].width;
                box.height = templ[0].height;
                boxes.push_back(box);
                scores.push_back(match.similarity);
            }
            cv_dnn::NMSBoxes(boxes, scores, 0, 0.5f, idxs);
            printf("matches %zu kept %zu\n", matches.size(), idxs.size());
            for (size_t i = 0; i < matches.size(); ++i) {
                uint32_t bits;
                memcpy(&bits, &matches[i].similarity, 4);
                printf("m %d %d %u %d %d %d\n", matches[i].x, matches[i].y, bits, matches[i].template_id, boxes[i].width, boxes[i].height);
            }
            for (int idx : idxs) printf("k %d\n", idx);
            return 0;
        }
        if (mode == "latency") {
            if (argc < 8) return usage();
            const std::string fmt = argv[2], class_id = argv[3], path = argv[4];
            const float threshold = (float)atof(argv[5]);
            const int num_features = atoi(argv[6]), n = atoi(argv[7]);
            const int pad = argc > 8 ? atoi(argv[8]) : 0;
            line2Dup::Detector detector(num_features, {4, 8});
            std::vector<std::string> ids{class_id};
            detector.readClasses(ids, fmt);
            Mat test_img = imread(path, IMREAD_UNCHANGED);
            if (test_img.empty()) { fprintf(stderr, "cannot read %s\n", path.c_str()); return 1; }
            Mat padded(test_img.rows + 2 * pad, test_img.cols + 2 * pad, test_img.type(), Scalar::all(0));
            test_img.copyTo(padded(Rect(pad, pad, test_img.cols, test_img.rows)));
            const int stride = 16;
            Mat img = padded(Rect(0, 0, stride * (padded.cols / stride), stride * (padded.rows / stride))).clone();
            size_t n_matches = 0;
            for (int i = 0; i < 10; ++i) n_matches = detector.match(img, threshold, ids).size();
            const auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < n; ++i) n_matches = detector.match(img, threshold, ids).size();
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
            printf("latency %.1f us per Detector::match, image %dx%dx%d, %d templates, %zu matches\n", us, img.rows, img.cols,
                   img.channels(), detector.numTemplates(), n_matches);
            // the same call with the caller's frame buffer page-locked (explicit: Detector::pinBuffer)
            detector.pinBuffer(img);
            for (int i = 0; i < 10; ++i) n_matches = detector.match(img, threshold, ids).size();
            const auto t1 = std::chrono::steady_clock::now();
            for (int i = 0; i < n; ++i) n_matches = detector.match(img, threshold, ids).size();
            const double us_pin = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t1).count() / n;
            printf("latency_pinned %.1f us per Detector::match, %zu matches\n", us_pin, n_matches);
            detector.unpinBuffer(img);
            // the throughput path from host memory: 32 frames per matchBatch, uploads overlapped with the kernels; the frames
            // live in one pinned block (a capture ring)
            const int nb = 32;
            Mat ring(img.rows * nb, img.cols, img.type());
            std::vector<Mat> frames;
            for (int b = 0; b < nb; ++b) {
                Mat f = ring(Rect(0, b * img.rows, img.cols, img.rows));
                img.copyTo(f);
                frames.push_back(f);
            }
            detector.pinBuffer(ring);
            size_t total = 0;
            for (int i = 0; i < 3; ++i) detector.matchBatch(frames, threshold, ids);
            const int reps = std::max(1, n / nb);
            const auto t2 = std::chrono::steady_clock::now();
            for (int i = 0; i < reps; ++i) {
                total = 0;
                for (const auto& l : detector.matchBatch(frames, threshold, ids)) total += l.size();
            }
            const double us_b = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t2).count() / (reps * nb);
            printf("batch_pinned %.1f us per frame (matchBatch of %d frames from pinned host memory), %zu matches per batch, PCIe bound %.1f us at 48 GB/s\n",
                   us_b, nb, total, (double)img.rows * img.cols * img.channels() / 48e3);
            detector.unpinBuffer(ring);
            return 0;
        }
        if (mode == "batch") {
            if (argc < 10) return usage();
            const std::string fmt = argv[2], class_id = argv[3], path = argv[4];
            const float threshold = (float)atof(argv[5]);
            const int num_features = atoi(argv[6]), nf = atoi(argv[7]), pad = atoi(argv[8]);
            std::vector<int> devices;
            for (const char* p = argv[9]; *p;) {
                devices.push_back(atoi(p));
                while (*p && *p != ',') ++p;
                if (*p == ',') ++p;
            }
            line2Dup::Detector detector(num_features, {4, 8});
            std::vector<std::string> ids{class_id};
            detector.readClasses(ids, fmt);
            Mat test_img = imread(path, IMREAD_UNCHANGED);
            if (test_img.empty()) { fprintf(stderr, "cannot read %s\n", path.c_str()); return 1; }
            Mat padded(test_img.rows + 2 * pad, test_img.cols + 2 * pad, test_img.type(), Scalar::all(0));
            test_img.copyTo(padded(Rect(pad, pad, test_img.cols, test_img.rows)));
            Mat img = padded(Rect(0, 0, 16 * (padded.cols / 16), 16 * (padded.rows / 16))).clone();
            // frame b = the image shifted 8 b columns to the right (wrapping)
            std::vector<Mat> frames;
            const int esz = img.channels();
            for (int b = 0; b < nf; ++b) {
                Mat f(img.rows, img.cols, img.type());
                const int sh = (8 * b) % img.cols;
                for (int y = 0; y < img.rows; ++y) {
                    memcpy(f.ptr(y) + (size_t)sh * esz, img.ptr(y), (size_t)(img.cols - sh) * esz);
                    memcpy(f.ptr(y), img.ptr(y) + (size_t)(img.cols - sh) * esz, (size_t)sh * esz);
                }
                frames.push_back(f);
            }
            auto same = [](const std::vector<line2Dup::Match>& a, const std::vector<line2Dup::Match>& b) {
                if (a.size() != b.size()) return false;
                for (size_t i = 0; i < a.size(); ++i)
                    if (!(a[i] == b[i]) || a[i].template_id != b[i].template_id) return false;
                return true;
            };
            std::vector<std::vector<line2Dup::Match>> single;
            for (const Mat& f : frames) single.push_back(detector.match(f, threshold, ids));
            const auto batch = detector.matchBatch(frames, threshold, ids);
            detector.matchAsync(frames, threshold, ids);
            const auto async = detector.wait();
            int ok_batch = batch.size() == single.size(), ok_async = async.size() == single.size();
            size_t total = 0;
            for (size_t f = 0; f < single.size(); ++f) {
                ok_batch = ok_batch && same(batch[f], single[f]);
                ok_async = ok_async && same(async[f], single[f]);
                total += single[f].size();
            }
            // a match() between matchAsync() and wait() on a detector limited to ONE concurrent call: the batch in flight keeps
            // its lane, the call in between gets another
            int ok_inter = 1;
            {
                line2Dup::Detector one(num_features, {4, 8});
                one.readClasses(ids, fmt);
                one.setConcurrency(1);
                one.matchAsync(frames, threshold, ids);
                ok_inter = same(one.match(frames[0], threshold, ids), single[0]);
                const auto w = one.wait();
                for (size_t f = 0; f < single.size(); ++f) ok_inter = ok_inter && w.size() == single.size() && same(w[f], single[f]);
            }
            // several contexts in one process: match() shards the templates, matchBatch() deals the frames
            line2Dup::Detector multi(num_features, {4, 8});
            multi.readClasses(ids, fmt);
            multi.setDevices(devices);
            int ok_dev = 1, ok_dev_batch = 1;
            for (size_t f = 0; f < frames.size(); ++f) ok_dev = ok_dev && same(multi.match(frames[f], threshold, ids), single[f]);
            const auto mb = multi.matchBatch(frames, threshold, ids);
            for (size_t f = 0; f < frames.size(); ++f) ok_dev_batch = ok_dev_batch && mb.size() == frames.size() && same(mb[f], single[f]);
            // unknown class: empty lists, as match() (:1136-1138)
            const auto none = detector.matchBatch(frames, threshold, {std::string("no_such_class")});
            int ok_none = none.size() == frames.size();
            for (const auto& l : none) ok_none = ok_none && l.empty();
            printf("batch frames %d matches %zu batch_same %d async_same %d devices %zu devices_same %d devices_batch_same %d unknown_class_empty %d async_interleaved_same %d\n", nf,
                   total, ok_batch, ok_async, devices.size(), ok_dev, ok_dev_batch, ok_none, ok_inter);
            for (const auto& m : single[0]) {
                uint32_t bits;
                memcpy(&bits, &m.similarity, 4);
                printf("%d %d %u %s %d\n", m.x, m.y, bits, m.class_id.c_str(), m.template_id);
            }
            return 0;
        }
        if (mode == "threads") {
            // Concurrent callers on ONE detector (the reference's match() is const and keeps no state, line2Dup.h:272-274):
            // n_threads host threads x n_calls match() calls each, two frame sizes interleaved, a batch thrown in now and then;
            // every list must equal the one the same call returns alone.  demo threads <fmt> <class> <image> <threshold>
            // <num_features> <n_threads> <n_calls> <pad> [max_lanes]
            if (argc < 10) return usage();
            const std::string fmt = argv[2], class_id = argv[3], path = argv[4];
            const float threshold = (float)atof(argv[5]);
            const int num_features = atoi(argv[6]), n_threads = atoi(argv[7]), n_calls = atoi(argv[8]), pad = atoi(argv[9]);
            line2Dup::Detector detector(num_features, {4, 8});
            if (argc > 10) detector.setConcurrency(atoi(argv[10]));
            std::vector<std::string> ids{class_id};
            detector.readClasses(ids, fmt);
            Mat test_img = imread(path, IMREAD_UNCHANGED);
            if (test_img.empty()) { fprintf(stderr, "cannot read %s\n", path.c_str()); return 1; }
            Mat padded(test_img.rows + 2 * pad, test_img.cols + 2 * pad, test_img.type(), Scalar::all(0));
            test_img.copyTo(padded(Rect(pad, pad, test_img.cols, test_img.rows)));
            // two geometries: the padded frame, and the same with 64 rows / 32 columns less
            std::vector<Mat> frames;
            frames.push_back(padded(Rect(0, 0, 16 * (padded.cols / 16), 16 * (padded.rows / 16))).clone());
            frames.push_back(padded(Rect(0, 0, 16 * (padded.cols / 16) - 32, 16 * (padded.rows / 16) - 64)).clone());
            auto same = [](const std::vector<line2Dup::Match>& a, const std::vector<line2Dup::Match>& b) {
                if (a.size() != b.size()) return false;
                for (size_t i = 0; i < a.size(); ++i)
                    if (!(a[i] == b[i]) || a[i].template_id != b[i].template_id) return false;
                return true;
            };
            std::vector<std::vector<line2Dup::Match>> alone;
            for (const Mat& f : frames) alone.push_back(detector.match(f, threshold, ids));
            std::vector<int> bad((size_t)n_threads, 0);
            std::vector<std::string> errs((size_t)n_threads);
            std::vector<std::thread> th;
            for (int t = 0; t < n_threads; ++t)
                th.emplace_back([&, t]() {
                    try {
                        for (int i = 0; i < n_calls; ++i) {
                            const size_t k = (size_t)((t + i) & 1);
                            if (i % 10 == 9) { // a batch of three frames of one size
                                const auto b = detector.matchBatch({frames[k], frames[k], frames[k]}, threshold, ids);
                                for (const auto& l : b)
                                    if (!same(l, alone[k])) ++bad[(size_t)t];
                            } else if (!same(detector.match(frames[k], threshold, ids), alone[k])) {
                                ++bad[(size_t)t];
                            }
                        }
                    } catch (const std::exception& e) {
                        errs[(size_t)t] = e.what();
                        ++bad[(size_t)t];
                    }
                });
            for (auto& t : th) t.join();
            int n_bad = 0;
            for (int t = 0; t < n_threads; ++t) {
                n_bad += bad[(size_t)t];
                if (!errs[(size_t)t].empty()) fprintf(stderr, "thread %d: %s\n", t, errs[(size_t)t].c_str());
            }
            printf("threads %d calls %d matches %zu %zu different %d\n", n_threads, n_calls, alone[0].size(), alone[1].size(), n_bad);
            return n_bad ? 2 : 0;
        }
        if (mode == "instance") {
            if (argc < 5) return usage();
            line2Dup::Detector* det = line2Dup::Detector::getInstance(argv[2]);
            line2Dup::Detector* again = line2Dup::Detector::getInstance(argv[2]); // the singleton is created once
            Mat img = imread(argv[3], IMREAD_UNCHANGED);
            if (img.empty()) { fprintf(stderr, "cannot read %s\n", argv[3]); return 1; }
            auto matches = det->match(img, (float)atof(argv[4]), det->classIds());
            printf("instance same %d classes %d templates %d T", det == again ? 1 : 0, det->numClasses(), det->numTemplates());
            for (int l = 0; l < det->pyramidLevels(); ++l) printf(" %d", det->getT(l));
            printf(" matches %zu\n", matches.size());
            for (const auto& m : matches) {
                uint32_t bits;
                memcpy(&bits, &m.similarity, 4);
                printf("%d %d %u %s %d\n", m.x, m.y, bits, m.class_id.c_str(), m.template_id);
            }
            return 0;
        }
        if (mode == "convert") { // readClasses + writeClasses: exercises the YAML subset without a GPU
            if (argc < 5) return usage();
            line2Dup::Detector detector(63, {4, 8});
            detector.readClasses({std::string(argv[3])}, argv[2]);
            detector.writeClasses(argv[4]);
            printf("converted %d templates\n", detector.numTemplates());
            return 0;
        }
        return usage();
    } catch (const std::exception& e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
}
